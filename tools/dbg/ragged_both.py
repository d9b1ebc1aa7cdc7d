"""Developer tool: both ragged real-shape lines of bench.py (independent and matched cluster sizes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
for sizes in (True, "matched"):
    r = bench.ragged_real_shape(dev, sizes=sizes)
    print(sizes, {k: r[k] for k in ("registrations_per_s", "ms_per_batch", "icp_iterations", "icp_kernel_ms_per_batch", "split_ms", "points_per_cluster_median")})
