"""Developer tool (-DICPFLOW_TAIL_CLOCK): serial / search clocks per iteration of the pairs of each association stage of the demo frame."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import _lib, frame_pairs, utils_match, utils_track
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ps, pd = G(g["point_src"]), G(g["point_dst"]); ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
a = frame_pairs.default_args(max_points=int(os.environ.get("MP", "10000")))
orig = utils_match._launch_pairs
def patched(args, st, dt, pairs):
    r = orig(args, st, dt, pairs)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 3072)(); _lib._L.icpflow_debug_tail_clock(buf)
    v = np.array(buf[:], dtype=np.int64).reshape(1024, 3)[:len(pairs)]
    its = np.maximum(v[:, 2], 1)
    n_src = st.h_count[r[0]]; n_dst = dt.h_count[r[1]]
    order = np.argsort(-(v[:, 0] + v[:, 1]))[:6]
    print(f"stage with {len(pairs)} pairs; the six with the most clocks (member 0 of a team):")
    for k in order:
        print(f"   pair {k}: {n_src[k]} x {n_dst[k]} points, {v[k, 2]} iterations, per iteration: serial part {v[k, 0] / its[k]:.0f} clocks, search + exchange {v[k, 1] / its[k]:.0f}")
    return r
utils_match._launch_pairs = patched
torch.manual_seed(0)
for _ in range(2):
    utils_track.track(a, ps, pd, ls, ld)
