"""Developer tool: shared window scans of the team kernel (round 5) against a library built with -DICPFLOW_NO_SHARE
(ICPFLOW_AB_LIB, default tools/dbg/libicpflow_prev.so): the transforms and iteration counts of team-shaped batches must be
bit-identical; prints the step times of both.  Each library runs in its own process (one HIP library per process)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [("ragged_matched_128x10000", 128, 10000, "matched", 20, 0, 100), ("ragged_independent_128x10000", 128, 10000, True, 20, 0, 100),
          ("teams_ragged_20x10000", 20, 10000, True, 500, 7, 50), ("teams_12x6000", 12, 6000, False, 20, 9, 50),
          ("ragged_matched_40x4096", 40, 4096, "matched", 200, 3, 100), ("ragged_matched_100x3000", 100, 3000, "matched", 100, 5, 100)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from types import SimpleNamespace
    from icp_flow_amd import _lib, synthetic, utils_match
    dev = torch.device("cuda:0")
    out = {}
    for name, B, N, ragged, nmin, seed, cap in SHAPES:
        S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=nmin)
        s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
        a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=cap, icp_stop_mode="reference")
        T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
        torch.cuda.synchronize()
        prof = _lib.Profile(64)
        reps = 10
        with _lib.options(profile=prof):
            t = time.perf_counter()
            for _ in range(reps): T2, it2 = utils_match.hist_icp(a, s, d, return_iterations=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / reps * 1e3
        icp, n = prof.collect()
        assert torch.equal(T, T2) and int(it) == int(it2)
        np.save(os.path.join(sys.argv[2], name + ".npy"), T.cpu().numpy())
        out[name] = dict(ms=round(ms, 3), icp_ms=round(icp / max(n, 1), 3), iters=int(it), finite=bool(torch.isfinite(T).all()))
    print("RESULT " + json.dumps(out))
    sys.exit(0)
import numpy as np, tempfile
res = {}
with tempfile.TemporaryDirectory() as d:
    for tag, lib in (("new", None), ("base", os.environ.get("ICPFLOW_AB_LIB", os.path.join(ROOT, "tools/dbg/libicpflow_prev.so")))):
        os.makedirs(os.path.join(d, tag))
        env = dict(os.environ)
        if lib: env["ICPFLOW_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(d, tag)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line: print(tag, "FAILED\n", r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
        res[tag] = json.loads(line[0][7:])
    ok = True
    for name, *_ in SHAPES:
        a, b = np.load(os.path.join(d, "new", name + ".npy")), np.load(os.path.join(d, "base", name + ".npy"))
        same = np.array_equal(a, b) and res["new"][name]["iters"] == res["base"][name]["iters"]
        ok = ok and same
        n, o = res["new"][name], res["base"][name]
        print(f"{name:32s} bit-identical {same}  iters {n['iters']:3d}/{o['iters']:3d}  step {o['ms']:.3f} -> {n['ms']:.3f} ms  icp {o['icp_ms']:.3f} -> {n['icp_ms']:.3f} ms  finite {n['finite']}"
              + ("" if same else f"  max|dT| {np.nanmax(np.abs(a - b)):.3e}, pairs differing {int((a != b).any(axis=(1, 2)).sum())}"))
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 2)
