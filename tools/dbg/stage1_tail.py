import ctypes, os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import _lib, utils_track, utils_match, frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
a = frame_pairs.default_args(max_points=10000); a.native_host = False   # (the hook below sits in the Python host)
kept = []
orig = utils_match._register_stage
def stash(args, st, dt, si, di, *rest):
    out = orig(args, st, dt, si, di, *rest)
    stage, scratch = out[1], out[2][3]
    clouds = scratch[: 2 * stage.K * stage.N * 4].view(2, stage.K, stage.N, 4)
    kept.append((clouds[0].clone(), clouds[1].clone(), st.h_count[si].copy(), dt.h_count[di].copy()))
    return out
utils_match._register_stage = stash
torch.manual_seed(0)
utils_track.track(a, ps, pd, ls, ld)
utils_match._register_stage = orig
S, D, cs, cd = kept[0]
for _ in range(2): utils_match.hist_icp_eval(a, S, D)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 3072)(); _lib._L.icpflow_debug_tail_clock(buf)
v = np.array(buf[:], dtype=np.int64).reshape(1024, 3)[:len(cs)]
its = np.maximum(v[:, 2], 1)
for k in np.argsort(-(v[:, 0] + v[:, 1]))[:14]:
    print(f"pair {k:3d}: {min(cs[k],10000):5d} x {min(cd[k],10000):5d} points, {v[k, 2]:3d} iterations, per iteration: serial {v[k, 0] / its[k]:7.0f}, search+exchange {v[k, 1] / its[k]:7.0f}, total {(v[k,0]+v[k,1])/2.4e6:.3f} ms")
print("iterations histogram:", np.bincount(np.minimum(v[:,2],100)//10).tolist())
if os.environ.get("SPLIT"):   # library built with -DICPFLOW_TAIL_CLOCK -DICPFLOW_TAIL_SPLIT: the phases of the pacing pairs
    sp = (ctypes.c_longlong * 16384)(); _lib._L.icpflow_debug_tail_split(sp)
    w = np.array(sp[:], dtype=np.int64).reshape(1024, 16)[:len(cs)]
    names = {1: "top of loop -> queries loaded", 11: "certificates + probes", 12: "window", 2: "scan", 10: "resolve + records", 3: "moments", 4: "block barrier",
             5: "totals + H", 13: "quartic coefficients", 14: "newton", 6: "adjugate + rotation", 15: "T, rmse", 9: "history + tally + stop check", 7: "cycle detection + publish"}
    for b in np.argsort(-(v[:, 0] + v[:, 1]))[:3]:
        print(f"pair {b} ({min(cs[b],10000)} x {min(cd[b],10000)}): clocks per iteration between the stamps of thread 0")
        for k in (1, 11, 12, 2, 10, 3, 4, 5, 13, 14, 6, 15, 9, 7):
            print(f"   {names[k]:34s} {w[b, k] / max(v[b, 2], 1):8.0f}")
