"""Registers, spills and LDS of the kernels in a built object (gfx950 code-object notes):
    python tools/kernel_resources.py icp [name-filter]
Extracts the device code of icp_flow_amd/csrc/_obj/<stem>.*.o with llvm-objdump --offloading and reads the notes."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
stem = sys.argv[1] if len(sys.argv) > 1 else "icp"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
objs = [o for o in glob.glob(os.path.join(ROOT, "icp_flow_amd", "csrc", "_obj", stem + ".*.o"))]
assert objs, "build first"
with tempfile.TemporaryDirectory() as d:
    o = os.path.join(d, "x.o")
    subprocess.check_call(["cp", objs[0], o])
    subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", o], stdout=subprocess.DEVNULL)
    co = [f for f in glob.glob(o + ".*") if "gfx950" in f][0]
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk)
    name = demangle(g("name").group(1))
    if flt and flt not in name:
        continue
    print(f"{name[:100]:100s} lds {g('group_segment_fixed_size').group(1):>6s} scratch {g('private_segment_fixed_size').group(1):>4s} "
          f"sgpr {g('sgpr_count').group(1):>3s} (spilled {g('sgpr_spill_count').group(1)}) vgpr {g('vgpr_count').group(1):>3s} (spilled {g('vgpr_spill_count').group(1)})")
