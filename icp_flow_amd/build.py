"""Build libicpflow_hip.so in-tree with hipcc for gfx950 (no torch, no cmake).

    python icp_flow_amd/build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU; the .so travels with the tree to the GPU box.  Every source is
compiled to its own object (in parallel, cached by content hash under csrc/_obj/) and linked once.
Flags that matter for parity:
  -ffp-contract=off                         every FMA in the kernels is an explicit fmaf()
  -fhip-fp32-correctly-rounded-divide-sqrt  IEEE division in the vote's bin index and
                                            correctly rounded sqrt of the NN distances
The library carries the hash of the sources it was built from (icpflow_build_info()); a library whose
hash differs from the tree's is stale and rebuilt, whatever the file times say.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(HERE, "libicpflow_hip.so")
SOURCES = ["api.hip", "hist.hip", "nn.hip", "icp.hip", "icp_fp32.hip", "pose.hip", "sort.hip", "cluster.hip", "hdbscan.hip", "table.hip", "assoc.hip", "frame.hip",
           "hdbscan_tree.cpp"]
HEADERS = ["common.hpp", "scan.hpp", "kernels.hpp", "kabsch.hpp", "posefuse.hpp", "votekey.hpp", "cluster_util.hpp", "sortdir.hpp",
           os.path.join("..", "..", "include", "icpflow_hip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(str(e).encode())
    return h.hexdigest()


def source_hash():
    """Hash of everything the library is built from (sources, headers, flags)."""
    return _digest([os.path.join(CSRC, f) for f in SOURCES + HEADERS], CFLAGS)[:16]


_MARK = b"ICPFLOW_SOURCE_HASH="


def built_hash(path=OUT):
    """The source hash baked into an existing library, or None.  Read from the file's bytes: loading the library
    here would pull the system HIP runtime into the process before torch brings its own (two runtimes in one
    process do not share devices)."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    k = blob.find(_MARK)
    if k < 0:
        return None
    tag = blob[k + len(_MARK): k + len(_MARK) + 16]
    return tag.decode() if len(tag) == 16 and all(c in b"0123456789abcdef" for c in tag) else None


def stale():
    return built_hash() != source_hash()


def _compile(src, extra):
    hdr = _digest([os.path.join(CSRC, f) for f in HEADERS], CFLAGS + list(extra))
    tag = _digest([os.path.join(CSRC, src)], [hdr])[:16]
    obj = os.path.join(OBJ, f"{os.path.splitext(src)[0]}.{tag}.o")
    if not os.path.exists(obj) or "-save-temps=obj" in extra:
        for old in os.listdir(OBJ):
            if old.startswith(os.path.splitext(src)[0] + ".") and old.endswith(".o"):
                os.remove(os.path.join(OBJ, old))
        subprocess.check_call([hipcc()] + CFLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj], cwd=CSRC)
    return obj


def build(force=False, extra=()):
    if not force and not stale():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    stamp = source_hash()
    jobs = {}
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        for src in SOURCES:
            # only api.hip sees the hash (the other objects stay cached when an unrelated file changes)
            e = list(extra) + ([f'-DICPFLOW_SOURCE_HASH="{stamp}"'] if src == "api.hip" else [])
            jobs[src] = ex.submit(_compile, src, e)
        objs = [jobs[s].result() for s in SOURCES]
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT + ".tmp"], cwd=CSRC)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    extra = ["-save-temps=obj"] if "--save-temps" in sys.argv else []
    print(build(force="--force" in sys.argv or bool(extra), extra=extra), source_hash())
