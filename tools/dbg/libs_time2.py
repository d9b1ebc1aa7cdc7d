"""Developer tool: step / ICP time of the headline shapes and the team shapes for several builds (LIBS=a.so,b.so; '' = product)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [("c2_256x1024", 256, 1024, False, 20, 0, 50, 30), ("c4shard_1024x2048", 1024, 2048, False, 20, 0, 50, 8), ("c4_8192x2048", 8192, 2048, False, 20, 0, 50, 3),
          ("ragged_matched", 128, 10000, "matched", 20, 0, 100, 10), ("ragged_indep", 128, 10000, True, 20, 0, 100, 10), ("teams_12x6000", 12, 6000, False, 20, 9, 50, 10)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from types import SimpleNamespace
    from icp_flow_amd import _lib, synthetic, utils_match
    dev = torch.device("cuda:0")
    out = []
    for name, B, N, ragged, nmin, seed, cap, reps in SHAPES:
        S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=nmin)
        s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
        a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=cap, icp_stop_mode="reference")
        T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
        torch.cuda.synchronize()
        prof = _lib.Profile(64)
        with _lib.options(profile=prof):
            t = time.perf_counter()
            for _ in range(reps): utils_match.hist_icp(a, s, d)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / reps * 1e3
        icp, n = prof.collect()
        out.append(f"{name} {ms:.3f}/{icp / max(n, 1):.3f}")
    print("RESULT " + " | ".join(out))
    sys.exit(0)
for lib in os.environ.get("LIBS", "").split(","):
    env = dict(os.environ)
    if lib: env["ICPFLOW_HIP_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"{(lib or 'product')[-16:]:16s}", line[0][7:] if line else "FAILED " + r.stderr[-1500:])
