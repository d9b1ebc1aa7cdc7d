"""Developer tool: bench.py's frame-pair measurement three times in one process (is the stream figure stable?)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
for k in range(3):
    r = bench.frame_pair_measurement(dev)
    print(k, {m: (r[m]["ms_per_frame_pair"], r[m]["stream_ms_per_frame_pair_4_in_flight"]) for m in ("max_points_2048", "max_points_10000")}, flush=True)
