"""Developer tool: phase times of the HDBSCAN host remainder (csrc/hdbscan_tree.cpp) on the demo frame pair's spanning
tree.  Compiles an instrumented copy of the source with g++ (timers at the numbered steps) and calls it through ctypes."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import utils_cluster
src = open(os.path.join(ROOT, "icp_flow_amd", "csrc", "hdbscan_tree.cpp")).read()
src = src.replace('#include "../../include/icpflow_hip.h"', '#include "%s/include/icpflow_hip.h"\n#include <chrono>\n#include <cstdio>\n'
                  'static double now(){return std::chrono::duration<double,std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();}' % ROOT)
marks = ["    // 1. orientation away from point 0", "    const auto byEnds =", "    // 2. single linkage: node n + i = i-th merge",
         "    // 3. condensed tree", "    // 4. stability per cluster id", "    // 6. labels: rank of the selected ids"]
for k, m in enumerate(marks):
    assert src.count(m) == 1, m
    src = src.replace(m, f"    double t{k}=now();\n" + m)
src = src.replace("    return 0;\n}\n", '    double t6=now();\n    fprintf(stderr,"orient %.2f sort %.2f linkage %.2f condense %.2f stability+eom %.2f labels %.2f total %.2f ms\\n",'
                  't1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5,t6-t0);\n    return 0;\n}\n')
d = tempfile.mkdtemp()
open(os.path.join(d, "t.cpp"), "w").write(src)
subprocess.check_call(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t.so")])
L = ctypes.CDLL(os.path.join(d, "t.so"))
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_demo.npz"))
pts = torch.from_numpy(np.concatenate([g["point_dst"], g["point_src"]], 0)).cuda()
t = utils_cluster.hdbscan_mst(pts, 21)
o = torch.argsort(t["w2"])
for name, idx in (("edges as they come", slice(None)), ("edges sorted by weight", o)):
    a = np.ascontiguousarray(t["a"][idx].cpu().numpy()); b = np.ascontiguousarray(t["b"][idx].cpu().numpy())
    w = np.ascontiguousarray(np.sqrt(t["w2"][idx].cpu().numpy()))
    lab = np.empty(len(pts), np.int32)
    print(name, file=sys.stderr)
    for _ in range(3):
        L.icpflow_hdbscan_labels(ctypes.c_void_p(a.ctypes.data), ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(w.ctypes.data),
                                 len(pts), 20, ctypes.c_void_p(lab.ctypes.data))
