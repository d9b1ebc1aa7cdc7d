import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from icp_flow_amd import frame_pairs, synthetic
from oracle import reference_path as rp
dev = torch.device("cuda", 0)
C = lambda x: torch.from_numpy(np.ascontiguousarray(x))
seed = 543
rng = np.random.default_rng(seed)
nobj = int(rng.integers(3, 14)); nmax = int(rng.choice([150, 400, 700])); mp = int(rng.choice([256, 512]))
d = synthetic.make_frame_pair(seed=seed, n_objects=nobj, n_max=nmax, n_background=int(rng.integers(200, 2500)))
fp = frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"])
a = frame_pairs.default_args(max_points=mp)
torch.manual_seed(0)
wp, wT = rp.match_pcds(a, C(fp.points_src), C(fp.points_dst), C(fp.labels_src), C(fp.labels_dst))
wflow = rp.flow_estimation_torch(C(fp.points_src), C(fp.labels_src), wp, wT, C(fp.pose)).numpy()
for name, native, assoc in (("native", True, None), ("python device", False, True), ("python host", False, False)):
    a.native_host, a.device_association = native, assoc
    out = frame_pairs.register_frame_pair(a, fp, dev)
    pairs = out["pairs"].cpu().numpy()
    err = np.linalg.norm(out["flow"].cpu().numpy() - wflow, axis=1)
    lab = fp.labels_src[np.argmax(err)]
    k = np.where(pairs[:, 0] == lab)[0]
    print(name, "association", out["association"], "worst flow difference %.3e m at a point of cluster %s" % (err.max(), lab), "pair row", pairs[k][:, :6] if len(k) else None, "oracle row", wp.numpy()[wp.numpy()[:, 0] == lab][:, :6])
