"""Developer tool: the host thread's busy time per frame pair on a stream of demo frame pairs (time inside the frame pairs'
generator steps, i.e. everything but waiting), device-side against host-side association, K in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import frame_pairs
g = load_golden("g8_demo"); lab = load_golden("g8_demo_labels")
dev = torch.device("cuda:0")
fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
orig = frame_pairs.register_frame_pair_steps
busy = [0.0]; steps = []
def timed(*a, **k):
    gen = orig(*a, **k)
    val = None; n = 0
    while True:
        t = time.perf_counter()
        try:
            y = gen.send(val)
        except StopIteration as done:
            d = time.perf_counter() - t; busy[0] += d; steps.append((n, d))
            return done.value
        d = time.perf_counter() - t; busy[0] += d; steps.append((n, d)); n += 1
        val = yield y
frame_pairs.register_frame_pair_steps = timed
if os.environ.get("PINNED") == "1":
    # the loader hands over pinned arrays: uploads become non-blocking copies
    pins = {}
    def upload(arr, device):
        key = arr.__array_interface__["data"][0]
        t = pins.get(key)
        if t is None:
            t = torch.from_numpy(arr).pin_memory(); pins[key] = t
        return t.to(device, non_blocking=True)
    frame_pairs._upload = upload
for mp in (2048, 10000):
    for mode in (True, False):
        a = frame_pairs.default_args(max_points=mp); a.device_association = mode
        for k in (4, 8):
            for _ in frame_pairs.register_in_flight_scheduler(a, [fp] * 8, dev, k): pass
            torch.cuda.synchronize(); busy[0] = 0.0; steps.clear(); t = time.perf_counter()
            n = 32
            for _ in frame_pairs.register_in_flight_scheduler(a, [fp] * n, dev, k): pass
            torch.cuda.synchronize(); wall = time.perf_counter() - t
            by = {}
            for s, d in steps: by.setdefault(s, []).append(d)
            print(f"{mp:5d} {'device' if mode else 'host  '} {k} in flight: {wall / n * 1e3:.3f} ms per frame pair, host busy {busy[0] / n * 1e3:.3f} ms "
                  f"(steps: {', '.join(f'{np.mean(v) * 1e3:.3f}' for _, v in sorted(by.items()))})", flush=True)
