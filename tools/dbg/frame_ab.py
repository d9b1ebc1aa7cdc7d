"""Developer tool: demo frame pair (ms, ICP launches) + the ragged real-shape batch with the library selected by ICPFLOW_HIP_LIB."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import _lib, utils_flow, utils_track, frame_pairs
import bench
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
eye = torch.eye(4, device=dev)
out = []
for mp in (10000, 2048):
    a = frame_pairs.default_args(max_points=mp)
    def run():
        torch.manual_seed(0)
        pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
        return utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, eye)
    for _ in range(3): ref = run()
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 5 * 1e3)
    prof = _lib.Profile(64)
    with _lib.options(profile=prof):
        for _ in range(4): run()
        torch.cuda.synchronize()
    icp, n = prof.collect()
    out.append(f"mp {mp}: {best:.3f} ms, ICP of both stages {icp / 4:.3f} ms ({n // 4} launches)")
r = bench.ragged_real_shape(dev)
out.append(f"ragged {r['registrations_per_s']:.0f}/s icp {r['icp_kernel_ms_per_batch']}")
print(os.path.basename(os.environ.get("ICPFLOW_HIP_LIB", "product")), " | ".join(out))
