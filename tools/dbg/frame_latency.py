"""Developer tool: where the latency of one demo frame pair goes -- every phase of match_pcds + flow bracketed by
device synchronisations (so the sum exceeds the unsynchronised figure printed first: the overlap is what is lost)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_flow, utils_track, utils_match, utils_check, frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000"))); a.native_host = False; a.device_association = os.environ.get("DEVICE_ASSOC", "0") == "1"   # (the phases below are those of the host path)
eye = torch.eye(4, device=dev)
def run():
    torch.manual_seed(0)
    pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
    return utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, eye)
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); print(f"unsynchronised: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per frame pair")
acc, calls = {}, {}
def timed(name, fn):
    def w(*args, **kw):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*args, **kw)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        h, d = acc.get(name, (0.0, 0.0)); acc[name] = (h + t1 - t0, d + t2 - t1); calls[name] = calls.get(name, 0) + 1
        return r
    return w
utils_check.ClusterTable.pair = staticmethod(timed("ClusterTable.pair", utils_check.ClusterTable.pair))
utils_match.ClusterTable.pair = utils_check.ClusterTable.pair
for mod, name in ((utils_match, "_sanity_mask"), (utils_match, "_register_stage"),
                  (utils_match, "sanity_grid"), (utils_match, "_finish_pairs"), (utils_match, "setdiff1d")):
    setattr(mod, name, timed(name, getattr(mod, name)))
utils_match.Pending.get = timed("Pending.get", utils_match.Pending.get)
utils_check.ClusterTable.fetch = timed("ClusterTable.fetch", utils_check.ClusterTable.fetch)
flow_fn = timed("flow_estimation_torch", utils_flow.flow_estimation_torch)
track = timed("track (whole)", utils_track.track)
R = 10
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(R):
    torch.manual_seed(0)
    pairs, Tm = track(a, ps, pd, ls, ld)
    flow_fn(a, ps, pd, ls, ld, pairs, Tm, eye)
torch.cuda.synchronize(); print(f"synchronised at every phase: {(time.perf_counter() - t) / R * 1e3:.3f} ms per frame pair")
print(f"{'phase':28s} calls   host ms   + wait for the device ms   (per frame pair)")
for k, (h, d) in acc.items():
    print(f"{k:28s} {calls[k] / R:5.1f}  {h / R * 1e3:8.3f}  {d / R * 1e3:8.3f}")
