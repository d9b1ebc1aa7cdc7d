"""CPU-only: the C-ABI library loads and exports every symbol include/icpflow_hip.h declares,
argument errors come back as status codes + messages (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(REPO, "include", "icpflow_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(icpflow_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as entry
    so = entry.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in icpflow_hip.h but not exported"


def test_python_binding_covers_the_header():
    from icp_flow_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    assert _lib.VERSION == 214
    assert re.fullmatch(r"[0-9a-f]{16}", _lib.BUILD_INFO), _lib.BUILD_INFO


def test_argument_errors_are_status_codes_with_messages():
    from icp_flow_amd import _lib
    L = _lib._L
    assert L.icpflow_workspace_bytes(0, 10, 0, 0, 0) == 0
    assert L.icpflow_workspace_bytes(256, 1024, 41, 41, 3) > 3 * 256 * 41 * 41 * 3 * 4
    rc = L.icpflow_hist_vote(None, None, 1, 1, 1, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1, 1, 1, None, None)
    assert rc == -1 and b"null pointer" in L.icpflow_last_error()
    one = ctypes.c_void_p(16)
    rc = L.icpflow_icp(one, one, None, 4, 8, 0.1, 5000, 1e-6, 0, None, None, None, None, None, one, 1 << 30, None, None)
    assert rc == -1 and b"max_iterations" in L.icpflow_last_error()
    rc = L.icpflow_icp(one, one, None, 4, 8, 0.1, 10, 1e-6, 7, None, None, None, None, None, one, 1 << 30, None, None)
    assert rc == -1 and b"stop_mode" in L.icpflow_last_error()
    rc = L.icpflow_icp(one, one, None, 4, 8, 0.1, 10, 1e-6, 0, None, None, None, None, None, one, 16, None, None)
    assert rc == -2 and b"workspace" in L.icpflow_last_error()
    rc = L.icpflow_nn_batch(one, one, 1, 4, 4, 2, 4, None, None, 1, one, one, None)
    assert rc == -1 and b"stride" in L.icpflow_last_error()
    # the stage entry points (version 206): structs checked before anything is launched
    tables, stage, reg = _lib.Tables(), _lib.Stage(), _lib.Registration()
    rc = L.icpflow_register_stage(None, ctypes.byref(stage), ctypes.byref(reg), one, 1 << 30, None, None)
    assert rc == -1 and b"null argument" in L.icpflow_last_error()
    rc = L.icpflow_register_stage(ctypes.byref(tables), ctypes.byref(stage), ctypes.byref(reg), one, 1 << 30, None, None)
    assert rc == -1 and b"null pointer" in L.icpflow_last_error()
    rc = L.icpflow_associate_frame(ctypes.byref(tables), ctypes.byref(stage), None, None, ctypes.byref(reg), 2.0, 0.2, 9.0, 0.2,
                                   one, 4, one, one, None, None, 0, None, None, one, 1 << 30, None, None)
    assert rc == -1 and b"null pointer" in L.icpflow_last_error()
    tables.d_table_src = tables.d_table_dst = 16
    stage.d_result = stage.d_si = stage.d_di = 16
    stage2 = _lib.Stage()
    stage2.K = 3
    rc = L.icpflow_associate_frame(ctypes.byref(tables), ctypes.byref(stage), ctypes.byref(stage2), None, ctypes.byref(reg), 2.0, 0.2,
                                   9.0, 0.2, one, 4, one, one, None, None, 0, None, None, one, 1 << 30, None, None)
    assert rc == -1 and b"stage 2 comes with" in L.icpflow_last_error()
    # one frame pair per call (version 207)
    par, pairs, need = _lib.FrameParams(), ctypes.c_int32(0), ctypes.c_size_t(0)
    rc = L.icpflow_track_frame(one, one, 10, one, one, 10, ctypes.byref(reg), ctypes.byref(par), one, one, ctypes.byref(pairs), None, None, None,
                               one, 1 << 30, ctypes.byref(need), None, None)
    assert rc == -1 and b"struct_size" in L.icpflow_last_error()
    rc = L.icpflow_track_frame(one, one, 10, one, one, 10, ctypes.byref(reg), None, one, one, ctypes.byref(pairs), None, None, None,
                               one, 1 << 30, ctypes.byref(need), None, None)
    assert rc == -1 and b"null pointer" in L.icpflow_last_error()
    par.struct_size, par.max_points = ctypes.sizeof(par), 2048
    rc = L.icpflow_track_frame(one, one, 10, one, one, 10, ctypes.byref(reg), ctypes.byref(par), one, one, ctypes.byref(pairs), None, None, None,
                               one, 16, ctypes.byref(need), None, None)
    assert rc == -2 and need.value > 16 and b"scratch" in L.icpflow_last_error()


def test_options_are_per_call_and_per_thread():
    """Tuning switches travel with each call (icpflow_options_t); the Python side keeps the options in force in a
    thread-local stack, so two host threads never see each other's settings."""
    import threading
    from icp_flow_amd import _lib
    assert _lib.opt() is None or _lib._DEFAULT_FLAGS                     # defaults -> NULL pointer
    seen = {}

    def other():
        seen["other"] = _lib._current()[-1]["search"]

    with _lib.options(search="grid", no_teams=True):
        o = ctypes.cast(_lib.opt(), ctypes.POINTER(_lib.Options)).contents
        assert (o.struct_size, o.icp_search, o.flags & _lib.OPT_FLAGS["no_teams"]) == (ctypes.sizeof(_lib.Options), 2, 64)
        with _lib.options(search="scan"):
            assert _lib._current()[-1]["search"] == 1 and _lib._current()[-1]["flags"] & 64
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen["other"] == 0
    assert _lib._current()[-1]["search"] == 0
    assert set(_lib.OPT_FLAGS.values()) == {1 << k for k in range(19)}   # seventeen bit-identity switches off + teams_half_gpu (bit 11) + two_launch (bit 17, opt-in)
    assert ctypes.sizeof(_lib.Options) == 96   # size_t, int, int, unsigned, pad, five pointers, two ints, two pointers on LP64
    # the structs of icpflow_register_stage / icpflow_associate_frame (include/icpflow_hip.h), LP64
    assert (ctypes.sizeof(_lib.Tables), ctypes.sizeof(_lib.Stage), ctypes.sizeof(_lib.Registration)) == (64, 56, 64)
    assert ctypes.sizeof(_lib.FrameParams) == 64        # icpflow_frame_params_t
    assert ctypes.sizeof(_lib.Mt19937) == 2500          # icpflow_mt19937_t


def test_product_refuses_cpu_tensors_no_fallback():
    from icp_flow_amd import hist, utils_helper, utils_match
    from types import SimpleNamespace
    x = torch.zeros(2, 8, 4)
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        utils_match.hist_icp(a, x, x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        hist.hist(x, x, -1, -1, -1, 1, 1, 1, 3, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        utils_helper.nearest_neighbor_batch(x, x)


def test_product_never_imports_the_oracle():
    """The oracle is the checker: nothing under icp_flow_amd/ may reference it."""
    pkg = os.path.join(REPO, "icp_flow_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f


def test_plain_c_program_consumes_the_header(tmp_path):
    """include/icpflow_hip.h is C (no C++/torch types): a C99 program compiled with gcc links the library, gets
    status codes for bad arguments and runs the host-side HDBSCAN tree function (no GPU needed)."""
    import subprocess
    import __graft_entry__ as entry
    so = entry.build()
    exe = str(tmp_path / "abi_consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "abi_consumer.c"), "-o", exe, so,
                           "-Wl,-rpath," + os.path.dirname(so)])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("ok ")


def test_per_call_buffers_are_not_inherited_by_nested_option_blocks():
    """options(vote_bins=..., icp_history=..., icp_init=..., icp_scale=...) name device buffers whose sizes follow ONE call;
    a nested block keeps the switches of the block around it but never its buffers."""
    from icp_flow_amd import _lib
    marker = object()
    with _lib.options(search="grid", no_teams=True, vote_bins=marker, icp_history=marker, icp_scale=marker, icp_init=(marker, marker)):
        cur = _lib._current()[-1]
        assert cur["vote_bins"] is marker and cur["icp_history"] is marker
        with _lib.options(no_score_prune=True):
            inner = _lib._current()[-1]
            assert inner["search"] == 2 and inner["flags"] & _lib.OPT_FLAGS["no_teams"] and inner["flags"] & _lib.OPT_FLAGS["no_score_prune"]
            assert inner["vote_bins"] is None and inner["icp_history"] is None and inner["icp_scale"] is None and inner["icp_init"] is None
        assert _lib._current()[-1]["vote_bins"] is marker
