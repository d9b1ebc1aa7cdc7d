"""Developer tool: the ragged real-shape batch of bench.py on its own (for rocprofv3 --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(bench.ragged_real_shape(torch.device("cuda:0")))
