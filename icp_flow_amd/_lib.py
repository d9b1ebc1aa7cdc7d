"""ctypes binding of libicpflow_hip.so (C ABI: include/icpflow_hip.h).

There is NO CPU fallback: if the HIP library is missing this module raises at
import, and every wrapper refuses non-GPU tensors.  torch is used only as the
owner of device memory and of the current HIP stream.
"""
import contextlib
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICPFLOW_HIP_LIB: developer override to load an instrumented build of the same ABI
LIB_PATH = os.environ.get("ICPFLOW_HIP_LIB") or os.path.join(_HERE, "libicpflow_hip.so")

STOP_REFERENCE = 0
STOP_PER_PAIR = 1

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python icp_flow_amd/build.py` "
        "(hipcc --offload-arch=gfx950).  icp_flow_amd has no CPU fallback.")

_L = ctypes.CDLL(LIB_PATH)

_f = ctypes.c_float
_d = ctypes.c_double
_i = ctypes.c_int
_p = ctypes.c_void_p
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/icpflow_hip.h declares
SIGNATURES = {
    "icpflow_version": (_i, []),
    "icpflow_last_error": (ctypes.c_char_p, []),
    "icpflow_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "icpflow_hist_vote": (_i, [_p, _p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, _p, _p]),
    "icpflow_hist_peaks": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "icpflow_nn_batch": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p, _p]),
    "icpflow_transform_points": (_i, [_p, _p, _i, _i, _p, _p]),
    "icpflow_count_valid": (_i, [_p, _i, _i, _p, _p]),
    "icpflow_build_info": (ctypes.c_char_p, []),
    "icpflow_estimate_init_pose": (_i, [_p, _p, _i, _i, _p, _i, _p, _i, _p, _i, _f, _p, _p, _sz, _p, _p]),
    "icpflow_icp": (_i, [_p, _p, _p, _i, _i, _d, _i, _d, _i, _p, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "icpflow_apply_icp": (_i, [_p, _p, _p, _i, _i, _d, _i, _d, _i, _p, _p, _p, _sz, _p, _p]),
    "icpflow_hist_icp": (_i, [_p, _p, _i, _i, _p, _i, _p, _i, _p, _i, _f, _d, _i, _d, _i, _p, _p, _p, _sz, _p, _p]),
    "icpflow_hist_icp_many": (_i, [_i, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i, _f, _d, _i, _d, _i, _p, _p, _p, _p, _p, _p]),
    "icpflow_match_eval": (_i, [_p, _p, _p, _i, _i, _d, _p, _p, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "icpflow_hist_icp_eval": (_i, [_p, _p, _i, _i, _p, _i, _p, _i, _p, _i, _f, _d, _i, _d, _i, _p, _p, _p, _p, _p, _p, _p,
                                   _p, _p, _sz, _p, _p]),
    "icpflow_gather_pad": (_i, [_p, _p, _i, _i, _p, _p]),
    "icpflow_gather_segments": (_i, [_p, _p, _p, _p, _i, _i, _p, _p]),
    "icpflow_assoc_assign": (_i, [_p, _p, _p, _i, _p, _i, _i, _f, _f, _f, _f, _p, _i, _p, _p, _p, _p, _p]),
    "icpflow_assoc_collect": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "icpflow_register_stage": (_i, [_p, _p, _p, _p, _sz, _p, _p]),
    "icpflow_associate_frame": (_i, [_p, _p, _p, _p, _p, _f, _f, _f, _f, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _sz, _p, _p]),
    "icpflow_register_stage_begin": (_i, [_p, _p, _p, _p, _sz, _p, _p, _p]),
    "icpflow_register_stage_finish": (_i, [_p, _p, _p, _p, _sz, _p, _p, _p]),
    "icpflow_associate_frame_begun": (_i, [_p, _p, _p, _p, _p, _f, _f, _f, _f, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _sz, _p, _p, _p]),
    "icpflow_track_frame": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p, _p, _p]),
    "icpflow_selftest_randperm": (_i, [_p, ctypes.c_int64, _i, _p]),
    "icpflow_cluster_stats": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "icpflow_cluster_table_workspace_bytes": (_sz, [_i, _i]),
    "icpflow_cluster_table": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _sz, _p]),
    "icpflow_cluster_table_pair_workspace_bytes": (_sz, [_i, _i, _i]),
    "icpflow_cluster_table_pair": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p, _sz, _p]),
    "icpflow_flow_rigid": (_i, [_p, _p, _i, _p, _p, _i, _p, _p, _p, _sz, _p]),
    "icpflow_flow_rigid_rows": (_i, [_p, _p, _i, _p, _i, _p, _i, _p, _p, _p]),
    "icpflow_dbscan_workspace_bytes": (_sz, [_i]),
    "icpflow_dbscan": (_i, [_p, _i, _p, _i, _d, _i, _p, _p, _p, _p, _sz, _p]),
    "icpflow_hdbscan_mst_workspace_bytes": (_sz, [_i]),
    "icpflow_hdbscan_mst": (_i, [_p, _i, _p, _i, _i, _d, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "icpflow_hdbscan_labels": (_i, [_p, _p, _p, _i, _i, _p]),
    "icpflow_selftest_vote_quotient": (_i, [_p, _i, _f, _f, _p, _p, _p]),
    "icpflow_profile_create": (_i, [_i, ctypes.POINTER(_p)]),
    "icpflow_profile_collect": (_i, [_p, ctypes.POINTER(_d), ctypes.POINTER(_i)]),
    "icpflow_profile_destroy": (_i, [_p]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(_L, _name)          # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args

VERSION = int(_L.icpflow_version())
BUILD_INFO = _L.icpflow_build_info().decode()


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with the library's message."""
    rc = getattr(_L, name)(*args)
    if rc != 0:
        msg = _L.icpflow_last_error()
        raise RuntimeError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")


SEARCH_AUTO, SEARCH_SCAN, SEARCH_GRID, SEARCH_SWEEP = 0, 1, 2, 3
ARITH_FP64, ARITH_FP32_REFERENCE = 0, 1
# developer switches (include/icpflow_hip.h ICPFLOW_OPT_*): each turns one optimisation off, results identical
OPT_FLAGS = {"no_sorted_vote": 1 << 0, "no_side_stream": 1 << 1, "no_eval_sweep": 1 << 2, "no_check_sweep": 1 << 3,
             "no_score_sweep": 1 << 4, "no_score_prune": 1 << 5, "no_teams": 1 << 6, "no_speculative": 1 << 7,
             "no_adaptive_windows": 1 << 8, "no_persistent": 1 << 9, "no_helpers": 1 << 10,
             # (not a bit-identity switch: teams on at most half of the CUs, two team launches side by side; icpflow_hip.h)
             "teams_half_gpu": 1 << 11, "no_shared_scans": 1 << 12,
             # icpflow_track_frame: stage 2's initial poses behind stage 1 instead of beside its ICP (same results; see icpflow_hip.h)
             "no_stage_overlap": 1 << 13, "no_vote_list": 1 << 14, "no_check_reuse": 1 << 15, "no_score_prebound": 1 << 16,
             # (opt-IN, bit-identical) ICP of a batch of a few rounds: the persistent grid drained for a second launch of whole-CU workgroups
             "two_launch": 1 << 17,
             # the sweeps' clouds sorted along the fixed cloud's longest axis only (no direction keys: csrc/sortdir.hpp)
             "no_dir_keys": 1 << 18}


class Options(ctypes.Structure):
    """icpflow_options_t: the per-call options of the fused entry points."""
    _fields_ = [("struct_size", _sz), ("icp_search", _i), ("icp_arith", _i), ("flags", ctypes.c_uint),
                ("profile", _p), ("d_vote_bins_u32", _p), ("d_icp_init_R", _p), ("d_icp_init_T", _p),
                ("d_icp_history", _p), ("icp_allow_reflection", _i), ("icp_estimate_scale", _i),
                ("d_icp_scale", _p), ("d_icp_init_s", _p), ("d_pair_active", _p)]


class Tables(ctypes.Structure):
    """icpflow_tables_t: both clouds of a frame pair as icpflow_cluster_table leaves them."""
    _fields_ = [("d_points_src", _p), ("d_order_src", _p), ("d_table_src", _p), ("d_points_dst", _p), ("d_order_dst", _p),
                ("d_table_dst", _p), ("S", _i), ("D", _i), ("label_stride", _i)]


class Stage(ctypes.Structure):
    """icpflow_stage_t: the candidate pairs of one association stage."""
    _fields_ = [("d_seg", _p), ("d_perm", _p), ("d_si", _p), ("d_di", _p), ("d_clouds", _p), ("d_result", _p), ("K", _i), ("N", _i)]


class Registration(ctypes.Structure):
    """icpflow_registration_t: the arguments of icpflow_hist_icp_eval that do not depend on the batch."""
    _fields_ = [("d_edges_x", _p), ("d_edges_y", _p), ("d_edges_z", _p), ("len_x", _i), ("len_y", _i), ("len_z", _i),
                ("decode_shift", _f), ("thres_dist", _d), ("relative_rmse_thr", _d), ("max_iterations", _i), ("stop_mode", _i)]


class Mt19937(ctypes.Structure):
    """icpflow_mt19937_t: the state of torch's CPU generator (at::mt19937)."""
    _fields_ = [("state", ctypes.c_uint32 * 624), ("index", ctypes.c_int32)]

    @staticmethod
    def from_torch(generator=None):
        """The engine state of a torch CPU generator (default: torch's global one) out of get_state() -- the legacy layout
        ATen keeps (CPUGeneratorImplStateLegacy: uint64 seed, int left, int seeded, uint64 next, uint64 state[624], ...)."""
        raw = (generator if generator is not None else torch.default_generator).get_state().numpy()
        if raw.size < 24 + 8 * 624:
            raise RuntimeError("unexpected layout of torch.Generator.get_state()")
        left = int(raw[8:12].view("<i4")[0])
        nxt = int(raw[16:24].view("<u8")[0])
        mt = Mt19937()
        ctypes.memmove(mt.state, raw[24:24 + 8 * 624].view("<u8").astype("<u4").tobytes(), 4 * 624)
        # at::mt19937 regenerates its block when --left reaches 0: left == 1 <=> every word of the block is consumed
        mt.index = 624 if left == 1 else nxt
        return mt

    def to_torch(self, generator=None):
        """Write the (advanced) state back into the torch generator it was taken from."""
        import numpy as np
        g = generator if generator is not None else torch.default_generator
        raw = g.get_state().numpy().copy()
        raw[24:24 + 8 * 624] = np.frombuffer(bytes(self.state), dtype="<u4").astype("<u8").view(np.uint8)
        idx = int(self.index)
        raw[8:12] = np.array([1 if idx >= 624 else 625 - idx], "<i4").view(np.uint8)
        raw[16:24] = np.array([idx], "<u8").view(np.uint8)
        g.set_state(torch.from_numpy(raw))


class FrameParams(ctypes.Structure):
    """icpflow_frame_params_t: the flags of the reference's parser that icpflow_track_frame reads."""
    _fields_ = [("struct_size", _sz), ("seed", ctypes.c_uint64), ("generator", _p), ("max_points", _i), ("min_cluster_size", _i),
                ("translation_frame", _f), ("thres_box", _f), ("thres_iou", _f), ("rot_limit_deg", _f), ("thres_error", _f),
                ("tight_padding", _i), ("superset_width", _i)]


class Profile:
    """icpflow_profile_t: HIP-event recorder of the ICP-iteration kernel's launches (owned by the caller)."""

    def __init__(self, capacity):
        self._h = _p()
        call("icpflow_profile_create", int(capacity), ctypes.byref(self._h))

    def collect(self):
        """-> (summed duration in ms, number of launches); re-arms the recorder."""
        ms, n = _d(0.0), _i(0)
        call("icpflow_profile_collect", self._h, ctypes.byref(ms), ctypes.byref(n))
        return float(ms.value), int(n.value)

    def close(self):
        if self._h:
            _L.icpflow_profile_destroy(self._h)
            self._h = _p()

    __del__ = close


_tls = threading.local()   # the options in force on THIS host thread (nothing process-global)


def _env_default_flags():
    """Developer convenience: ICPFLOW_<SWITCH>=1 in the environment turns that optimisation off for every call
    that does not say otherwise (parsed here, on the Python side; the library itself reads no environment)."""
    f = 0
    for name, bit in OPT_FLAGS.items():
        if os.environ.get("ICPFLOW_" + name.upper(), "0") not in ("0", ""):
            f |= bit
    return f


_DEFAULT_FLAGS = _env_default_flags()


def _current():
    return getattr(_tls, "stack", None) or [dict(search=0, arith=0, flags=_DEFAULT_FLAGS, profile=None, vote_bins=None,
                                                 icp_init=None, icp_history=None, icp_allow_reflection=False,
                                                 icp_scale=None, pair_active=None)]


@contextlib.contextmanager
def options(search=None, arith=None, profile=None, vote_bins=None, icp_init=None, icp_history=None,
            icp_allow_reflection=None, icp_scale=None, pair_active=None, **switches):
    """Per-call options for every wrapper invoked inside the `with` block on this thread.
    search: 'auto' | 'scan' | 'grid' | 'sweep';  arith: 'fp64' | 'fp32_reference';  profile: a Profile;
    vote_bins: uint32 device tensor [B, Lx*Ly*Lz] receiving the fused vote's bins;  switches: no_teams=True ..."""
    cur = dict(_current()[-1])
    # device buffers belong to ONE call (their sizes follow that call's B, max_iterations, histogram): a nested block never
    # inherits them from the block around it -- it names them itself or runs without
    cur.update(vote_bins=None, icp_init=None, icp_history=None, icp_scale=None, pair_active=None)
    if search is not None:
        cur["search"] = {"auto": 0, "scan": 1, "grid": 2, "sweep": 3}.get(search, search)
    if arith is not None:
        cur["arith"] = {"fp64": 0, "fp32_reference": 1}.get(arith, arith)
    if profile is not None:
        cur["profile"] = profile
    if vote_bins is not None:
        cur["vote_bins"] = vote_bins
    if icp_init is not None:        # (R [B,3,3], T [B,3]) float32 device tensors: init_transform of icpflow_icp
        cur["icp_init"] = icp_init
    if icp_history is not None:     # float32 [max_iterations, B, 16] device tensor: t_history of icpflow_icp
        cur["icp_history"] = icp_history
    if icp_allow_reflection is not None:
        cur["icp_allow_reflection"] = bool(icp_allow_reflection)
    if icp_scale is not None:       # float32 [B] device tensor: estimate_scale of icpflow_icp, receives s
        cur["icp_scale"] = icp_scale
    if pair_active is not None:     # uint8 [B] device tensor: pairs flagged 0 are not in the batch (hist_icp / hist_icp_eval)
        cur["pair_active"] = pair_active
    for k, v in switches.items():
        cur["flags"] = (cur["flags"] | OPT_FLAGS[k]) if v else (cur["flags"] & ~OPT_FLAGS[k])
    stack = getattr(_tls, "stack", None)
    if stack is None:
        stack = _tls.stack = []
    stack.append(cur)
    try:
        yield
    finally:
        stack.pop()


def opt():
    """ctypes pointer to the icpflow_options_t in force on this thread (NULL = library defaults)."""
    cur = _current()[-1]
    if (cur["search"] == 0 and cur["arith"] == 0 and cur["flags"] == 0 and cur["profile"] is None
            and cur["vote_bins"] is None and cur["icp_init"] is None and cur["icp_history"] is None
            and not cur["icp_allow_reflection"] and cur["icp_scale"] is None and cur.get("pair_active") is None):
        return None
    dp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None   # noqa: E731
    # (ICPFLOW_OPTIONS_SIZE: developer override for loading an older build of the ABI, whose struct was a prefix of this one)
    o = Options(int(os.environ.get("ICPFLOW_OPTIONS_SIZE", ctypes.sizeof(Options))), int(cur["search"]), int(cur["arith"]), int(cur["flags"]),
                cur["profile"]._h if cur["profile"] is not None else None, dp(cur["vote_bins"]),
                dp(cur["icp_init"][0] if cur["icp_init"] is not None else None),
                dp(cur["icp_init"][1] if cur["icp_init"] is not None else None), dp(cur["icp_history"]),
                1 if cur["icp_allow_reflection"] else 0, 1 if cur["icp_scale"] is not None else 0, dp(cur["icp_scale"]),
                dp(cur["icp_init"][2] if cur["icp_init"] is not None and len(cur["icp_init"]) > 2 else None),
                dp(cur.get("pair_active")))
    _tls.last = o          # keep the struct alive until this thread builds the next one
    return ctypes.cast(ctypes.pointer(o), _p)


def workspace_bytes(B, N, lens=(0, 0, 0)):
    return int(_L.icpflow_workspace_bytes(int(B), int(N), int(lens[0]), int(lens[1]), int(lens[2])))


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("icp_flow_amd: input must be a GPU (HIP) tensor -- there is no CPU path "
                               "(the reference refuses CPU tensors too, hist_cuda/cpp/hist.cpp:22)")


def cloud(t, name="cloud"):
    """Validate a [B,N,4] float32 cloud; returns it contiguous."""
    require_gpu(t)
    if t.dim() != 3 or t.shape[2] != 4:
        raise RuntimeError(f"{name}: expected shape [B,N,4] (x,y,z,flag), got {tuple(t.shape)}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(device):
    """The current stream of `device` as an integer handle.  torch.cuda.current_stream() builds a Stream object through
    several Python layers (7.5 us a call, 22 calls per frame pair: a tenth of the host thread's time on a stream of frame
    pairs, tools/dbg/stream_host_profile.py); torch's raw getter is one C call."""
    if _raw_stream is not None:
        idx = device.index if isinstance(device, torch.device) else None
        if idx is None:
            idx = torch.cuda.current_device()
        return int(_raw_stream(idx))
    return int(torch.cuda.current_stream(device).cuda_stream)


def stream(device):
    return ctypes.c_void_p(stream_handle(device))


_ws_cache = {}


def workspace(device, nbytes):
    """A cached, grow-only scratch buffer per (device, current stream): calls issued on different streams
    may overlap on the GPU and must not share scratch (torch caching-allocator owned)."""
    key = (device.type, device.index, stream_handle(device))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def workspaces(device, sizes):
    """Distinct cached scratch buffers for the batches of one icpflow_hist_icp_many call (slot k of the current stream)."""
    out = []
    for k, nbytes in enumerate(sizes):
        key = (device.type, device.index, stream_handle(device), "many", k)
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            _ws_cache[key] = buf
        out.append(buf)
    return out


def check_vote_bins(B, lens):
    """The debug export of the fused vote (options(vote_bins=...)) is written without size information on the C side:
    refuse a tensor that is not uint32-sized [B, Lx*Ly*Lz] on the device."""
    t = _current()[-1]["vote_bins"]
    if t is None:
        return
    need = int(B) * int(lens[0]) * int(lens[1]) * int(lens[2])
    if not t.is_cuda or t.element_size() != 4 or t.numel() < need or not t.is_contiguous():
        raise RuntimeError(f"options(vote_bins=...): need a contiguous 4-byte device tensor of at least {need} elements "
                           f"([B, Lx*Ly*Lz] = [{B}, {need // max(int(B), 1)}]), got {tuple(t.shape)} {t.dtype}")
