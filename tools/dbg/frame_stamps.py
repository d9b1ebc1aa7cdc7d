"""Developer tool: where the HOST's time goes inside icpflow_track_frame on the demo frame pair (one at a time, inputs resident):
medians of the intervals between the library's time stamps (icpflow_debug_frame_stamps), at max_points 2048 and 10000."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import _lib, frame_pairs
dev = torch.device("cuda:0")
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
ego = torch.eye(4, device=dev)
names = ["enqueue tables", "generator blocks", "wait for tables", "candidates", "segments + draws", "enqueue stage 1", "enqueue the rest", "wait for matches"]
buf = (ctypes.c_double * 9)()
sub = (ctypes.c_double * 5)()
for mp in (2048, 10000):
    a = frame_pairs.default_args(max_points=mp)
    for _ in range(3): frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
    rows, tot, subs = [], [], []
    for _ in range(21):
        torch.cuda.synchronize(); t = time.perf_counter()
        frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps)
        torch.cuda.synchronize(); tot.append((time.perf_counter() - t) * 1e3)
        _lib._L.icpflow_debug_frame_stamps(buf)
        v = np.array(buf[:]); rows.append(np.diff(v))
        _lib._L.icpflow_debug_frame_substamps(sub); subs.append(np.diff(np.array(sub[:])))
    med = np.median(np.array(rows), axis=0)
    print(f"max_points {mp}: {np.median(tot):.3f} ms per frame pair; inside the call {med.sum():.0f} us: " + ", ".join(f"{n} {m:.0f}" for n, m in zip(names, med)))
    sm = np.median(np.array(subs), axis=0)
    print(f"   segments + draws = stage 2's superset {sm[0]:.0f}, workspace sizes {sm[1]:.0f}, stage 1's segments and draws {sm[2]:.0f}, stage 2's segments and index lists {sm[3]:.0f}")
