import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_flow, utils_track
from oracle import reference_path as rp
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
for mp in [int(x) for x in os.environ.get("MP", "2048,10000").split(",")]:
    a = rp.default_args(max_points=mp, min_cluster_size=20, translation_frame=2.0, thres_box=0.1, thres_rot=0.1, thres_error=0.2, thres_iou=0.2)
    def run():
        torch.manual_seed(0)
        pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
        flow = utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, torch.eye(4, device=dev))
        return pairs, Tm, flow
    pairs, Tm, flow = run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
    err = np.linalg.norm(flow.cpu().numpy() - g["flow"], axis=1)
    epe = float(np.linalg.norm(flow.cpu().numpy() - g["gt_flow"], axis=1).mean())
    print(f"max_points {mp}: {len(pairs)} pairs, {ms:.2f} ms / frame pair, flow within 1e-4 of ref(2048): {np.mean(err<1e-4):.4f}, max {err.max():.3e}, EPE {epe:.4f}")
