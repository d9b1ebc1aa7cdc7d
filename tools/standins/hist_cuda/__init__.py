"""Shadows /root/reference/hist_cuda (whose hist.py JIT-builds a CUDA extension at
import time, hist.py:36-37) with a CPU restatement of the vote kernel."""
