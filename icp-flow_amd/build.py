"""Build libicpflow_hip.so in-tree with hipcc for gfx950 (no torch, no cmake).

    python icp-flow_amd/build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU; the .so travels with the tree to the GPU box.
Flags that matter for parity:
  -ffp-contract=off                         every FMA in the kernels is an explicit fmaf()
  -fhip-fp32-correctly-rounded-divide-sqrt  IEEE division in the vote's bin index and
                                            correctly rounded sqrt of the NN distances
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libicpflow_hip.so")
SOURCES = ["api.hip", "hist.hip", "nn.hip", "icp.hip", "pose.hip", "sort.hip", "cluster.hip", "hdbscan.hip", "hdbscan_tree.cpp"]
HEADERS = ["common.hpp", "scan.hpp", "kernels.hpp", "votekey.hpp", "cluster_util.hpp", os.path.join("..", "..", "include", "icpflow_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra=()):
    if not force and not stale():
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra) + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", OUT]
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    extra = ["-save-temps=obj"] if "--save-temps" in sys.argv else []
    print(build(force="--force" in sys.argv or bool(extra), extra=extra))
