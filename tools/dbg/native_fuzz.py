"""Developer check: random labelled synthetic frame pairs (ragged clusters, relabelled objects, over-long clusters that get
subsampled, small max_points that force the fall-back) through icpflow_track_frame and through the Python host with the
device-side association (which falls back to its host-side association where the superset falls short; the native call
registers the exact stage 2 itself): bit for bit."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import frame_pairs, synthetic
dev = torch.device("cuda", 0)
bad = served = unserved = 0
first, trials = int(os.environ.get("FIRST", "100")), int(os.environ.get("TRIALS", "60"))
for seed in range(first, first + trials):
    rng = np.random.default_rng(seed)
    nobj = int(rng.integers(2, 16)); nmax = int(rng.choice([120, 300, 700, 1500, 3000])); mp = int(rng.choice([128, 256, 512, 2048, 10000]))
    d = synthetic.make_frame_pair(seed=seed, n_objects=nobj, n_max=nmax, n_background=int(rng.integers(100, 4000)))
    fp = frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"])
    a = frame_pairs.default_args(max_points=mp, tight_padding=bool(rng.integers(0, 2)))
    a.native_host, a.device_association = False, True
    want = frame_pairs.register_frame_pair(a, fp, dev)
    got = frame_pairs.register_frame_pair_native(a, fp, dev)
    torch.cuda.synchronize()
    if got is None:
        # (until the call registered the reference's stage 2 alone itself: no cluster keeps its label and passes the sanity
        # check, i.e. stage 1 has no candidate -- checked here on the Python host's tables; no longer expected)
        from icp_flow_amd.utils_check import ClusterTable, _sanity_mask
        G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        st, dt = ClusterTable.pair(G(fp.points_src), G(fp.labels_src).float(), G(fp.points_dst), G(fp.labels_dst).float())
        lab = np.unique(np.concatenate([st.h_labels.astype(np.int64), dt.h_labels.astype(np.int64)]))
        pr = np.stack([lab, lab], 1)[lab >= 0].astype(np.float32)
        a.translation_frame = want["translation_frame"]
        ok = len(pr) == 0 or not _sanity_mask(a, st, dt, pr).any()
        unserved += 1
    else:
        ok = frame_pairs._served(got) and all(torch.equal(got[k], want[k]) for k in ("pairs", "transformations", "flow"))
    served += want["association"] == "device"
    bad += not ok
    print(f"seed {seed}: objects {nobj} n_max {nmax} max_points {mp} tight {a.tight_padding}: matched {len(want['pairs'])}, "
          f"python host's association: {want['association']}{'' if ok else '   <-- look'}")
print(f"frame pairs {trials}, of which the superset served {served} (the others: exact stage 2 inside the call; {unserved} without a "
      f"stage-1 candidate left to the finer-grained path), different: {bad}")
