"""Developer tool: bench.py's frame-pair extras (one at a time, 4 and 8 in flight) with the association on the device and on the
host (frame_pairs.default_args is patched), alternating, on one box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from icp_flow_amd import frame_pairs
dev = torch.device("cuda:0")
orig = frame_pairs.default_args
for rep in range(int(os.environ.get("REPS", "2"))):
    for mode in (True, False):
        def patched(*a, **k):
            r = orig(*a, **k); r.device_association = mode; return r
        frame_pairs.default_args = patched
        res = bench.frame_pair_measurement(dev, clustering=False) if "clustering" in bench.frame_pair_measurement.__code__.co_varnames else bench.frame_pair_measurement(dev)
        frame_pairs.default_args = orig
        for mp in ("max_points_2048", "max_points_10000"):
            e = res[mp]
            print("device" if mode else "host  ", mp, e["ms_per_frame_pair"], "| 4 in flight", e["stream_ms_per_frame_pair_4_in_flight_passes"],
                  "| 8 in flight", e["stream_ms_per_frame_pair_8_in_flight_passes"], flush=True)
