"""Developer tool: the demo frame pair through the device-side association (args.device_association, the default) and through the
host path: same matched pairs, flow difference, latency of both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs, utils_flow, utils_track
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
eye = torch.eye(4, device=dev)
for mp in (2048, 10000):
    res = {}
    for mode in (False, True, False, True):
        a = frame_pairs.default_args(max_points=mp); a.native_host = False   # (device against host association of the PYTHON host)
        a.device_association = mode
        a.generator = torch.Generator()
        def run():
            a.generator.manual_seed(0)
            pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
            return pairs, utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, eye)
        for _ in range(3): pairs, flow = run()
        ts = []
        for _ in range(11):
            torch.cuda.synchronize(); t = time.perf_counter(); pairs, flow = run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        res[mode] = (pairs.cpu().numpy(), flow.cpu().numpy())
        print(f"max_points {mp} {'device' if mode else 'host  '} association: {len(pairs)} pairs, median {sorted(ts)[5]:.3f} ms, min {min(ts):.3f}")
    ph, fh = res[False]; pd_, fd = res[True]
    same = ph.shape == pd_.shape and np.array_equal(ph[:, :2], pd_[:, :2])
    print(f"   same matched pairs (in order): {same}; flow: max difference {np.abs(fh - fd).max():.3e} m; rows max difference {np.abs(ph - pd_).max() if same else float('nan'):.3e}")
    ref = load_golden("g8_demo_cudatopk" if mp == 2048 else "g8_demo_mp10000_cudatopk")
    print(f"   device path against the reference run: {np.abs(fd - ref['flow']).max():.3e} m")
