import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd.utils_check import ClusterTable
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
for _ in range(3): ClusterTable.pair(ps, ls, pd, ld)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): ClusterTable.pair(ps, ls, pd, ld)
torch.cuda.synchronize(); print("ClusterTable.pair ms", (time.perf_counter() - t) / 20 * 1e3)
t = time.perf_counter()
for _ in range(20): o1 = torch.argsort(ls, stable=True); o2 = torch.argsort(ld, stable=True)
torch.cuda.synchronize(); print("two argsorts ms", (time.perf_counter() - t) / 20 * 1e3)
