"""ctypes binding of oracle_core.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  The product package (icp_flow_amd/) must never do so.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_core.so")
_lib = None


def build(force=False):
    """Compile oracle_core.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle_core.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(src)):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_hist_vote.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int] \
            + [ctypes.c_float] * 6 + [ctypes.c_int] * 3 + [f32p]
        L.oracle_hist_vote.restype = None
        L.oracle_knn1.argtypes = [f32p, f32p] + [ctypes.c_int] * 5 + [i64p, i64p, i64p, f32p, f32p]
        L.oracle_knn1.restype = None
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _f32(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)) if a is not None else None


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def hist_vote(X, Y, mins, maxs, lens):
    """X, Y: float32 [B,N,4] torch/numpy; returns torch float32 [B,Lx,Ly,Lz]."""
    Xn = np.ascontiguousarray(torch.as_tensor(X).detach().cpu().numpy(), dtype=np.float32)
    Yn = np.ascontiguousarray(torch.as_tensor(Y).detach().cpu().numpy(), dtype=np.float32)
    assert Xn.ndim == 3 and Yn.ndim == 3 and Xn.shape[2] == 4 and Yn.shape[2] == 4
    assert Xn.shape[0] == Yn.shape[0]
    B, NX, NY = Xn.shape[0], Xn.shape[1], Yn.shape[1]
    lx, ly, lz = (int(v) for v in lens)
    out = np.empty((B, lx, ly, lz), dtype=np.float32)
    # float(t) of a float32 0-dim tensor is exact; c_float conversion then
    # restores the original float32 value (the pybind `const float` argument of
    # hist.cpp:5-7 does the same).
    mn = [float(v) for v in mins]
    mx = [float(v) for v in maxs]
    lib().oracle_hist_vote(_f32(Xn), _f32(Yn), B, NX, NY, mn[0], mn[1], mn[2],
                           mx[0], mx[1], mx[2], lx, ly, lz, _f32(out))
    return torch.from_numpy(out)


def knn1(p1, p2, lengths1=None, lengths2=None, return_nn=False):
    """p1 [B,N1,>=3], p2 [B,N2,>=3] float32 -> (d2 [B,N1], idx int64 [B,N1], nn or None)."""
    a = np.ascontiguousarray(torch.as_tensor(p1).detach().cpu().numpy(), dtype=np.float32)
    b = np.ascontiguousarray(torch.as_tensor(p2).detach().cpu().numpy(), dtype=np.float32)
    assert a.ndim == 3 and b.ndim == 3 and a.shape[0] == b.shape[0]
    B, N1, s1 = a.shape
    _, N2, s2 = b.shape
    l1 = None if lengths1 is None else np.ascontiguousarray(
        torch.as_tensor(lengths1).cpu().numpy(), dtype=np.int64)
    l2 = None if lengths2 is None else np.ascontiguousarray(
        torch.as_tensor(lengths2).cpu().numpy(), dtype=np.int64)
    idx = np.empty((B, N1), dtype=np.int64)
    d2 = np.empty((B, N1), dtype=np.float32)
    nn = np.empty((B, N1, 3), dtype=np.float32) if return_nn else None
    lib().oracle_knn1(_f32(a), _f32(b), B, N1, N2, s1, s2, _i64(l1), _i64(l2),
                      _i64(idx), _f32(d2), _f32(nn) if return_nn else None)
    return (torch.from_numpy(d2), torch.from_numpy(idx),
            torch.from_numpy(nn) if return_nn else None)
