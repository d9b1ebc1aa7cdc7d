"""Developer tool: four config-2 batches through ONE call (icpflow_hist_icp_many) against one after the other, for several builds
(LIBS=a.so,b.so; '' = the product library), each in a process of its own."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from types import SimpleNamespace
    from icp_flow_amd import synthetic, utils_match
    dev = torch.device("cuda:0")
    B, N = 256, 1024
    a = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=50, icp_stop_mode="reference")
    many = [synthetic.make_batch(B, N, seed=0, first=k * B) for k in range(4)]
    srcs = [torch.from_numpy(m[0]).to(dev) for m in many]
    dsts = [torch.from_numpy(m[1]).to(dev) for m in many]
    for _ in range(3): utils_match.hist_icp_many(a, srcs, dsts)
    res = []
    for fn in (lambda: utils_match.hist_icp_many(a, srcs, dsts), lambda: [utils_match.hist_icp(a, s, d) for s, d in zip(srcs, dsts)]):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 10 * 1e3)
        res.append(best)
    print("RESULT many %.3f ms (%.0f k/s) | one after the other %.3f ms (%.0f k/s)" % (res[0], 4 * B / res[0], res[1], 4 * B / res[1]))
    sys.exit(0)
for lib in os.environ.get("LIBS", "").split(","):
    env = dict(os.environ)
    if lib: env["ICPFLOW_HIP_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"{(lib or 'product')[-16:]:16s}", line[0][7:] if line else "FAILED " + r.stderr[-1500:])
