"""Developer check: labelled synthetic frame pairs (ragged clusters, relabelled objects, over-long clusters that get
subsampled) through track() + flow against the oracle's match_pcds / flow, many seeds."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import frame_pairs, synthetic
from oracle import reference_path as rp
dev = torch.device("cuda", 0)
C = lambda x: torch.from_numpy(np.ascontiguousarray(x))
bad = 0
for seed in range(int(os.environ.get("FIRST", "10")), int(os.environ.get("FIRST", "10")) + int(os.environ.get("TRIALS", "12"))):
    rng = np.random.default_rng(seed)
    nobj = int(rng.integers(3, 14)); nmax = int(rng.choice([150, 400, 700])); mp = int(rng.choice([256, 512]))
    d = synthetic.make_frame_pair(seed=seed, n_objects=nobj, n_max=nmax, n_background=int(rng.integers(200, 2500)))
    fp = frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"])
    a = frame_pairs.default_args(max_points=mp)
    out = frame_pairs.register_frame_pair(a, fp, dev)
    torch.manual_seed(0)
    wp, wT = rp.match_pcds(a, C(fp.points_src), C(fp.points_dst), C(fp.labels_src), C(fp.labels_dst))
    wflow = rp.flow_estimation_torch(C(fp.points_src), C(fp.labels_src), wp, wT, C(fp.pose)).numpy()
    pairs = out["pairs"].cpu().numpy()
    same_pairs = pairs.shape == tuple(wp.shape) and np.array_equal(pairs[:, 0:2], wp.numpy()[:, 0:2])
    err = np.linalg.norm(out["flow"].cpu().numpy() - wflow, axis=1).max() if same_pairs else float("nan")
    flag = "" if (same_pairs and err < 1e-4) else "   <-- look"
    bad += flag != ""
    print(f"seed {seed}: objects {nobj} n_max {nmax} max_points {mp}: matched {len(pairs)} / oracle {len(wp)}, "
          f"same pairs {same_pairs}, worst flow difference {err:.2e} m{flag}")
print("frames with differences:", bad)
