"""Developer check: DBSCAN labels and HDBSCAN spanning trees against the oracle on many random clouds."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_flow_amd import utils_cluster
from oracle import cluster as oc, hdbscan as oh
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
bad = 0
for trial in range(int(os.environ.get("TRIALS", "60"))):
    n = int(rng.integers(2, 2500))
    k = int(rng.integers(2, 10))
    centers = rng.uniform(-6, 6, size=(k, 3)) * np.array([1, 1, 0.3])
    sig = rng.uniform(0.05, 0.6)
    p = (centers[rng.integers(0, k, n)] + rng.normal(0, sig, size=(n, 3))).astype(np.float32)
    if rng.random() < 0.3:
        p = np.round(p * 8) / 8           # heavy ties
    p = p.astype(np.float32)
    eps = float(rng.choice([0.1, 0.25, 0.4, 1.0])); mp = int(rng.integers(1, 12))
    mask = rng.random(n) < 0.85 if rng.random() < 0.5 else None
    lab = utils_cluster.dbscan(p, eps, mp, mask)[0].cpu().numpy()
    want = np.full(n, -2, np.int64)
    sel = np.ones(n, bool) if mask is None else mask
    want[sel] = oc.dbscan_components(p[sel], eps, mp)
    ok1 = np.array_equal(lab, want)
    ms = int(rng.integers(1, 9))
    ok2 = True
    if sel.sum() > ms + 1:
        t = utils_cluster.hdbscan_mst(p, ms, mask)
        a, b = t["a"].cpu().numpy().astype(np.int64), t["b"].cpu().numpy().astype(np.int64)
        lo, hi = np.minimum(a, b), np.maximum(a, b); o = np.lexsort((hi, lo))
        ra, rb, rw, rc = oh.mst(p[sel], ms)
        rows = np.flatnonzero(sel)
        ok2 = np.array_equal(lo[o], rows[ra]) and np.array_equal(hi[o], rows[rb]) and np.array_equal(t["w2"].cpu().numpy()[o], rw)
    if not (ok1 and ok2):
        bad += 1
        print("MISMATCH trial", trial, n, eps, mp, ms, ok1, ok2)
print("trials done, mismatches:", bad)
