"""Developer tool: bench.py's frame_pair_measurement on its own (no other extras before it)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
r = bench.frame_pair_measurement(dev)
for mp in ("max_points_2048", "max_points_10000"):
    f = r[mp]
    print(mp, "native", f["ms_per_frame_pair"], f["ms_per_frame_pair_runs"], "python host", f["ms_per_frame_pair_python_host"], "track+flow", f["ms_per_frame_pair_track_then_flow"],
          "stream4", f["stream_ms_per_frame_pair_4_in_flight"])
