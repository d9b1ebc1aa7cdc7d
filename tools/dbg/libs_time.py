"""Developer tool: step / ICP time of team-shaped batches for several builds of the library (LIBS=a.so,b.so; '' = product), each
in its own process."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SHAPES = [("ragged_matched_128x10000", 128, 10000, "matched", 20, 0, 100), ("ragged_independent_128x10000", 128, 10000, True, 20, 0, 100),
          ("teams_ragged_20x10000", 20, 10000, True, 500, 7, 50), ("teams_12x6000", 12, 6000, False, 20, 9, 50),
          ("ragged_matched_40x4096", 40, 4096, "matched", 200, 3, 100), ("ragged_matched_100x3000", 100, 3000, "matched", 100, 5, 100)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from types import SimpleNamespace
    from icp_flow_amd import _lib, synthetic, utils_match
    dev = torch.device("cuda:0")
    out = []
    for name, B, N, ragged, nmin, seed, cap in SHAPES:
        S, D, _ = synthetic.make_batch(B, N, seed=seed, ragged=ragged, n_min=nmin)
        s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
        a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=cap, icp_stop_mode="reference")
        T, it = utils_match.hist_icp(a, s, d, return_iterations=True)
        torch.cuda.synchronize()
        prof = _lib.Profile(64)
        with _lib.options(profile=prof):
            t = time.perf_counter()
            for _ in range(10): utils_match.hist_icp(a, s, d)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 10 * 1e3
        icp, n = prof.collect()
        out.append(f"{name} {ms:.3f}/{icp / max(n, 1):.3f} it {int(it)}")
    print("RESULT " + " | ".join(out))
    sys.exit(0)
for lib in os.environ.get("LIBS", "").split(","):
    env = dict(os.environ)
    if lib: env["ICPFLOW_HIP_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"{lib or 'product':36s}", line[0][7:] if line else "FAILED " + r.stderr[-1500:])
