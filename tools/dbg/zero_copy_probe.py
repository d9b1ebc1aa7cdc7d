"""Developer tool: what an upload of a stage's few KB costs the host thread -- pageable .to(), pinned non-blocking .to(), and no
upload at all (the kernel reads the rows from pinned host memory) -- and that the last gives the same clouds."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icp_flow_amd import _lib
dev = torch.device("cuda:0")
M, B, N = 60000, 97, 2048
pts = torch.randn(M, 3, device=dev)
order = torch.randperm(M, device=dev)
rng = np.random.default_rng(0)
seg = np.empty((3, B), np.int64); seg[0] = rng.integers(0, M - N, B); seg[1] = rng.integers(20, N, B); seg[2] = -1
perm_n = 20000
host = np.concatenate([seg.reshape(-1).view(np.uint8), rng.integers(0, 1000, perm_n).astype(np.int32).view(np.uint8)])
pinned = torch.empty(len(host), dtype=torch.uint8, pin_memory=True); pinned.numpy()[:] = host
out = torch.empty((B, N, 4), device=dev)
def gather(ptr):
    _lib.call("icpflow_gather_segments", _lib.ptr(pts), _lib.ptr(order), ctypes.c_void_p(ptr), None, B, N, _lib.ptr(out), _lib.stream(dev))
def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    h = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize(); w = (time.perf_counter() - t) / n * 1e6
    return h, w
def pageable():
    d = torch.from_numpy(host).to(dev); gather(d.data_ptr())
def pinned_copy():
    d = pinned.to(dev, non_blocking=True); gather(d.data_ptr())
def zero_copy():
    gather(pinned.data_ptr())
for name, fn in (("pageable .to + gather", pageable), ("pinned non-blocking .to + gather", pinned_copy), ("gather reads pinned host memory", zero_copy)):
    h, w = timed(fn)
    print(f"{name:36s} host {h:6.1f} us per call, with the GPU drained {w:6.1f} us")
pageable(); a = out.clone(); out.zero_(); zero_copy(); torch.cuda.synchronize()
print("same clouds:", bool(torch.equal(a, out)))
small = np.zeros(16, np.float32)
h, w = timed(lambda: torch.from_numpy(small).to(dev))
print(f"pageable .to of 64 B: host {h:.1f} us")
