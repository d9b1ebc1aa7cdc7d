"""bench.py -- cluster-pair ICP registrations / second on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full registration (`hist_icp`: translation-histogram vote, NMS + top-5,
6-candidate scoring, <= 50 ICP iterations with the reference's batch-global stop, roll-back
check) of every pair of the workload, inputs resident in HBM before the timed region.

  --gpus 1  (headline) BASELINE config 2: 256 cluster pairs x 1024 points.
  --gpus N  (N > 1)    BASELINE config 4: 8192 cluster pairs x 2048 points in total, contiguous blocks of
            8192/N pairs per rank (sharding.shard_range; pair k is synthetic pair k on every N), STRONG scaling;
            every step ends with the exchange the sharded product issues: match_eval fills the 40-byte
            pair row, and ONE RCCL all_gather of [B,26] rows (transform + pair row, sharding.pack_rows) runs inside the
            timed region -- the path has no other exchange step.  The single-GPU line carries the same workload on
            one GPU (`config4_on_one_gpu`) so that the scaling curve has its N = 1 point.
  --workload config2|config4 overrides the choice.
  --force-collective   world of ONE rank: initialise RCCL anyway and run the all_gather (single-GPU boxes exercise the
            N > 1 code path); the gathered rows are compared with the local ones after the timed region (`gather_check`).
  With no launcher around it (WORLD_SIZE unset) `--gpus N > 1` starts its N ranks itself (self_launch: the contract's
  own torch.distributed.run command line on 127.0.0.1 and a free port); fewer than N devices => "needs N devices, found M".

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import icp_flow_amd  # noqa: E402,F401  (first: the package asks for sixteen hardware queues before the HIP runtime initialises, icp_flow_amd/__init__.py)
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_LANE_OPS = 78.6e12    # 256 CU x 128 fp32 lanes x 2.4 GHz; 157.3 TFLOP/s counting fma = 2
LANE_OPS_PER_EVAL = 8           # one 3-D squared distance + compare = 8 lane-ops
COUNTERS_FILE = os.path.join(REPO, "profiles", "r06_icp_kernel_counters.json")
RAGGED_COUNTERS_FILE = os.path.join(REPO, "profiles", "r06_ragged_counters.json")
CONFIG4_COUNTERS_FILE = os.path.join(REPO, "profiles", "r06_config4_shard_counters.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="auto", choices=["auto", "config2", "config4", "stream"],
                    help="stream: frame pairs dealt round-robin to the ranks (BASELINE configs 3 / 5's shape), ms / frame pair")
    ap.add_argument("--frames", default=None, help="stream: directory of frame-pair / sequence .npz files (default: the reference's "
                                                   "demo frame pair from tests/golden, --frame-pairs copies of it)")
    ap.add_argument("--frame-pairs", type=int, default=None, help="stream: frame pairs per step IN TOTAL (default 64; the first so many of --frames)")
    ap.add_argument("--in-flight", type=int, default=4, help="stream: frame pairs in flight per GPU")
    ap.add_argument("--max-points", type=int, default=10000, help="stream: the reference's max_points (demo.sh: 10000)")
    ap.add_argument("--pairs", type=int, default=None, help="cluster pairs per step IN TOTAL (config2: 256 per GPU)")
    ap.add_argument("--points", type=int, default=None, help="points per cluster (= padded length)")
    ap.add_argument("--iters", type=int, default=50, help="ICP iteration cap (BASELINE: 50)")
    ap.add_argument("--stop-mode", default="reference", choices=["reference", "per_pair"])
    ap.add_argument("--cpu-pairs", type=int, default=256, help="pairs in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--time-every", type=int, default=5, help="HIP events around the ICP launch in every n-th step of the timed region (1 = every step)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements")
    ap.add_argument("--force-collective", action="store_true", help="one rank: init RCCL and all_gather anyway")
    ap.add_argument("--no-n1", action="store_true", help="N > 1: skip the single-GPU run of the same workload on rank 0 (n1_same_workload)")
    ap.add_argument("--check-gather", action="store_true", help="compare the gathered rows with the local ones")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="no GPU work: the ranks run the launcher, the rendezvous, the sharding and the one all_gather of "
                         "[B,26] rows over gloo on CPU tensors (what tests/test_bench_launcher.py drives)")
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here, the way the
    contract's own command line does -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py <the same arguments>` -- and hand its exit code on.  Rank 0 of the child world
    prints the one JSON line on the stdout this process shares with it."""
    import subprocess
    have = a.gpus if a.launch_selftest else torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} needs {a.gpus} devices, found {have}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_selftest(a, rank, world):
    """The N > 1 plumbing of main() without a GPU: environment of the launcher, gloo rendezvous on 127.0.0.1, contiguous
    shards of the config-4 pair list, `steps` exchanges of the [B,26] rows through the ONE all_gather of the path
    (sharding.gather_results with the counts every rank knows), max over ranks of the timed region, one JSON line from
    rank 0.  The rows are made up (a generator seeded by the pair index stands in for hist_icp_eval): this checks the
    launcher and the exchange, it measures nothing."""
    import torch.distributed as dist
    from icp_flow_amd.sharding import gather_results, pack_rows, shard_range, unpack_rows
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = a.pairs or 8192
    counts = [shard_range(r, world, total)[1] for r in range(world)]
    first, B = shard_range(rank, world, total)
    g = torch.Generator().manual_seed(2024)
    allrows = torch.randn(total, 24, generator=g)
    mine = allrows[first:first + B]
    rows = pack_rows(mine[:, :16].reshape(B, 4, 4).contiguous(), first, mine[:, 16:18], mine[:, 18:20], mine[:, 20:22], mine[:, 22:24])
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = gather_results(rows, world, counts=counts, force_collective=True)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    Tg, pr = unpack_rows(got)
    ok = bool(torch.equal(Tg.reshape(total, 16), allrows[:, :16]) and torch.equal(pr[:, 2:], allrows[:, 16:]) and
              torch.equal(pr[:, 0], torch.arange(total, dtype=torch.float32)))
    if rank == 0:
        print(json.dumps({"launch_selftest": True, "metric": "cluster-pair ICP registrations/sec", "value": None,
                          "unit": "registrations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(float(tmax.item()) / max(a.steps, 1) * 1e3, 4), "scaling": "strong",
                          "data": "made-up rows (no GPU work)",
                          "config": {"pairs_total": total, "pairs_per_gpu": counts},
                          "gather_check": {"rows": list(got.shape), "identical_to_all_ranks_rows": ok,
                                           "backend": dist.get_backend(), "rccl_library": None},
                          # (the fields every N > 1 line carries: main() fills them from a run of the same step on rank 0's GPU alone)
                          "n1_same_workload": {"value": None, "unit": "registrations/s", "pairs": total, "what": "no GPU work in the launch selftest"},
                          "scaling_efficiency": None}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit("launch selftest: the gathered rows differ from the rows the ranks contributed")


def stream_frames(a, selftest):
    """The stream's frame pairs (the same list on every rank): --frames DIR, else copies of the reference's demo frame pair
    (tests/golden/g8_demo*.npz; BASELINE config 1's data, the only real frame in the tree); --launch-selftest: small synthetic
    labelled frame pairs (no GPU work is done on them).  -> (list of paths or FramePair objects, description)"""
    from icp_flow_amd import frame_pairs, synthetic
    total = a.frame_pairs or 64
    if selftest:
        fps = []
        for k in range(total):
            d = synthetic.make_frame_pair(seed=k, n_objects=3, n_max=60, n_background=50)
            fps.append(frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"], d["gt_flow"], name=f"synthetic-{k}"))
        return fps, f"{total} small synthetic frame pairs (launch selftest, no registration)"
    if a.frames:
        paths = frame_pairs.list_frame_pairs(a.frames)
        if not paths:
            raise SystemExit(f"bench.py --workload stream: no .npz under {a.frames}")
        if a.frame_pairs:
            paths = paths[:a.frame_pairs]
        return paths, f"{len(paths)} files under {a.frames}"
    gdir = os.path.join(REPO, "tests", "golden")
    g, lab = np.load(os.path.join(gdir, "g8_demo.npz")), np.load(os.path.join(gdir, "g8_demo_labels.npz"))
    fp = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"], name="demo.npz")
    return [fp] * total, f"{total} copies of the reference's demo.npz frame pair (63 k points per frame, labels of the G8 fixture)"


def stream_main(a, rank, world, local):
    """`--workload stream` (SURVEY 8(e) second half, BASELINE.json's "ms/frame-pair"): the frame pairs are dealt round-robin to
    the ranks (frame_pairs.shard_round_robin: a frame pair's association stays on one GPU), every rank keeps `--in-flight` of its
    share in flight (one icpflow_track_frame call per frame pair), and ONE all_reduce per step adds up the counts -- the only
    collective of the path.  A step = one pass over the whole stream; inputs resident in HBM; the timed region is
    bracketed by barrier + synchronize, the max over ranks is reported.  Accuracy (EPE against the frames' ground truth)
    comes from an untimed pass through frame_pairs.run_stream (its own all_reduce of the weighted sums).
    --launch-selftest: the same launcher, sharding and collectives over gloo with a stand-in for the registration."""
    import torch.distributed as dist
    from icp_flow_amd import frame_pairs
    selftest = a.launch_selftest
    collective = world > 1 or a.force_collective
    if selftest:
        dev = torch.device("cpu")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        collective = True
    else:
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} needs device {local}, found {torch.cuda.device_count()} device(s)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if collective:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    items, what = stream_frames(a, selftest)
    args = frame_pairs.default_args(max_points=a.max_points)
    mine = frame_pairs.shard_round_robin(items, rank, world)
    counts = [len(frame_pairs.shard_round_robin(items, r, world)) for r in range(world)]
    if selftest:
        def stand_in(args_, fp, device):      # (what a registration returns, made up from the frame pair alone)
            k = int(len(fp.points_src)) % 7 + 1
            return dict(pairs=torch.zeros((k, 10)), transformations=torch.eye(4).repeat(k, 1, 1), flow=torch.from_numpy(fp.gt_flow.copy()))
        fps = list(mine)
    else:
        fps = [fp for it in mine for fp in ([it] if isinstance(it, frame_pairs.FramePair) else frame_pairs.load_any(it, args))]
        seen = {}
        for fp in fps:                        # inputs resident in HBM (copies of one frame pair share one upload)
            if id(fp) not in seen:
                seen[id(fp)] = frame_pairs.make_resident(fp, dev)

    def one_pass():
        n = matched = 0
        if selftest:
            for fp in fps:
                matched += int(stand_in(args, fp, dev)["pairs"].shape[0]); n += 1
        else:
            for _, fp, out in frame_pairs.register_in_flight(args, fps, dev, a.in_flight):
                matched += int(out["pairs"].shape[0]); n += 1
        tot = torch.tensor([n, matched], dtype=torch.float64, device=dev)
        if collective:
            dist.all_reduce(tot)              # the one collective of a step
        return tot

    def sync():
        if collective:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        one_pass()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tot = one_pass()
    sync()
    dt = time.perf_counter() - t0
    if collective:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    n_total, matched_total = int(tot[0].item()), int(tot[1].item())
    # the N = 1 point of this line: rank 0 runs the WHOLE stream alone (same in-flight depth, no collective), two passes after a warm-up
    n1 = None
    if world > 1 and rank == 0 and not a.no_n1:
        try:
            if selftest:
                t1 = time.perf_counter(); k1 = sum(1 for _ in items); d1 = max(time.perf_counter() - t1, 1e-9); reps = 1
            else:
                everything = [fp for it in items for fp in ([it] if isinstance(it, frame_pairs.FramePair) else frame_pairs.load_any(it, args))]
                for fp in everything:
                    if id(fp) not in seen:
                        seen[id(fp)] = frame_pairs.make_resident(fp, dev)
                run_all = lambda: sum(1 for _ in frame_pairs.register_in_flight(args, everything, dev, a.in_flight))   # noqa: E731
                run_all(); torch.cuda.synchronize(dev)
                reps = 2
                t1 = time.perf_counter()
                for _ in range(reps):
                    k1 = run_all()
                torch.cuda.synchronize(dev)
                d1 = time.perf_counter() - t1
            n1 = {"value": round(d1 * 1e3 / max(k1 * reps, 1), 4), "unit": "ms/frame-pair", "frame_pairs": k1, "passes": reps,
                  "what": "the whole stream on rank 0's GPU alone, after the timed region; scaling_efficiency = n1 / (n_gpus * value)"}
        except Exception as e:
            n1 = {"error": repr(e)}
    # accuracy, untimed: the stream through run_stream (EPE & co. against the frames' ground truth, its one all_reduce of the sums)
    summary = frame_pairs.run_stream(args, items, dev, rank, world, register_fn=stand_in if selftest else None,
                                     in_flight=1 if selftest else a.in_flight)
    if rank == 0:
        lib = None if selftest else next((ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln), None)
        from icp_flow_amd import _lib
        out = {"metric": "ms/frame-pair (frame-pair stream)", "value": round(dt * 1e3 / max(n_total * a.steps, 1), 4),
               "unit": "ms/frame-pair", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / max(a.steps, 1) * 1e3, 4), "higher_is_better": False, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "made-up results (no GPU work)" if selftest else ("files" if a.frames else "the reference's demo.npz frame pair, replicated"),
               "frame_pairs_per_s": round(n_total * a.steps / dt, 2),
               "config": {"workload": f"frame-pair stream: {what}; match_pcds + per-point flow per frame pair (icpflow_track_frame), "
                                      f"max_points {a.max_points}, {a.in_flight} frame pairs in flight per GPU, dealt round-robin to {world} rank(s)",
                          "frame_pairs_total": len(items), "frame_pairs_per_gpu": counts, "in_flight": a.in_flight, "max_points": a.max_points,
                          "sharding": "round-robin over ranks; ONE all_reduce of (frame pairs, matched cluster pairs) per step inside the timed region"
                                      if collective else "one rank, no exchange step"},
               "matched_cluster_pairs_per_step": matched_total,
               "accuracy": {k: summary[k] for k in ("frame_pairs", "matched_cluster_pairs", "evaluated_points", "epe", "accs", "accr", "outlier", "Routlier", "pose_sources") if k in summary},
               "reduce_check": {"frame_pairs_counted_by_all_ranks": n_total, "equals_sum_of_shares": n_total == sum(counts) == summary["frame_pairs"],
                                "backend": dist.get_backend() if collective else None, "rccl_library": lib},
               "library_build": None if selftest else _lib.BUILD_INFO}
        if selftest:
            out["launch_selftest"] = True
        if world > 1:
            out["n1_same_workload"] = n1
            out["scaling_efficiency"] = (round(n1["value"] / (world * out["value"]), 4) if n1 and n1.get("value") and out["value"] else None)
        print(json.dumps(out))
    if collective:
        dist.destroy_process_group()
    if not (n_total == sum(counts) == summary["frame_pairs"]):
        raise SystemExit("stream: the ranks' frame pairs do not add up to the stream")


def timed_steps(step, sync, steps, warmup, iters_cap, every=1):
    """W untimed steps, then exactly K steps between two (barrier + device synchronize); the launches of the
    dominant kernel are bracketed by HIP events on the stream they run on (icpflow_profile_t) -- in every `every`-th
    step of the timed region: an event pair costs the step ~7 us of a 0.68 ms chain of dependent launches
    (tools/dbg/profile_cost_ab.py), so the headline times one step in five.  -> ..., steps whose launches were timed"""
    from icp_flow_amd import _lib
    for _ in range(warmup):
        step()
    # one event pair per ICP launch: one launch per step up to 128 iterations (speculative single launch), one per
    # iteration beyond
    prof = _lib.Profile(steps * (iters_cap if iters_cap > 128 else 1) + 8)
    every = max(1, int(every))
    timed = 0
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        if k % every == 0:
            with _lib.options(profile=prof):
                T, iters = step()
            timed += 1
        else:
            T, iters = step()
    sync()
    dt = time.perf_counter() - t0
    icp_ms, icp_launches = prof.collect()
    prof.close()
    return dt, icp_ms, icp_launches, T, iters, timed


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.force_collective or a.launch_selftest):
        self_launch(a)                      # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world                          # under a launcher the world it made is the truth
    if a.workload == "stream":
        return stream_main(a, rank, world, local)
    if a.launch_selftest:
        return launch_selftest(a, rank, world)
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} needs device {local}, found {torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    collective = world > 1 or a.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from icp_flow_amd import _lib, synthetic, utils_match
    from icp_flow_amd.sharding import gather_results, pack_rows, shard_range, unpack_rows

    workload = a.workload if a.workload != "auto" else ("config2" if world == 1 else "config4")
    if workload == "config2":        # weak: every rank its own 256 pairs (only ever run with --gpus 1 by default)
        N = a.points or 1024
        total = (a.pairs or 256) * world
        scaling = "weak"
    else:                            # strong: the same 8192 pairs whatever N
        N = a.points or 2048
        total = a.pairs or 8192
        scaling = "strong"
    counts = [shard_range(r, world, total)[1] for r in range(world)]
    first, B = shard_range(rank, world, total)                  # contiguous block of pairs
    S, D, _ = synthetic.make_batch(B, N, seed=0, first=first)   # pair k = default_rng(k): the same pairs on every N
    src = torch.from_numpy(S).to(dev)
    dst = torch.from_numpy(D).to(dev)
    from types import SimpleNamespace
    args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N,
                           icp_max_iterations=a.iters, icp_stop_mode=a.stop_mode)

    def step():
        if not collective:
            return utils_match.hist_icp(args, src, dst, return_iterations=True)
        # what the sharded product exchanges: transform + the 40-byte pair row, ONE RCCL all_gather
        # over xGMI of [B,26] float32 rows, no host sync (every rank knows all counts from shard_range); hist_icp +
        # match_eval as the one call match_pairs makes (icpflow_hist_icp_eval)
        T, ev, iters = utils_match.hist_icp_eval(args, src, dst, return_iterations=True)
        rows = pack_rows(T, first, ev[0], ev[1], ev[2], ev[3])
        return gather_results(rows, world, counts=counts, force_collective=True), iters

    def sync():
        if collective:
            dist.barrier()
        torch.cuda.synchronize(dev)

    dt, icp_ms, icp_launches, T, iters, steps_timed = timed_steps(step, sync, a.steps, a.warmup, a.iters, every=(a.time_every if a.iters <= 128 else 1))
    gather_check = None
    if collective:
        if True:                # (always, outside the timed region) the rows this rank contributed, as every rank received them
            Tl, ev, _ = utils_match.hist_icp_eval(args, src, dst, return_iterations=True)
            mine = pack_rows(Tl, first, ev[0], ev[1], ev[2], ev[3])
            gather_check = {"rows": list(T.shape), "identical_to_local_rows": bool(torch.equal(T[first:first + B], mine)),
                            "pair_index_column_ok": bool(torch.equal(T[:, 16], torch.arange(total, device=dev, dtype=torch.float32))),
                            "backend": dist.get_backend(),
                            "rccl_library": next((ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln), None)}
        T = unpack_rows(T)[0]
    if collective:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    iters_done = int(iters.item())
    if iters_done < 0:
        raise SystemExit("hist_icp abandoned a batch (team timeout): the measurement is void")

    if rank != 0:
        if collective:
            dist.destroy_process_group()
        return

    value = total * a.steps / dt
    # The N = 1 point of THIS line's curve: with more than one rank, rank 0 times the SAME step -- hist_icp +
    # match_eval over all `total` pairs as one batch on one GPU, less the all_gather -- after the timed region, with the same
    # protocol (the other ranks have left; nothing here is part of `value`).  A driver that divides this line's value by the
    # `--gpus 1` line's would divide config 4 by config 2: `scaling_efficiency` is the figure that means something.
    n1 = None
    if collective and workload == "config4" and not a.no_n1:   # (--force-collective: a world of one exercises it too)
        try:
            n1 = same_workload_on_one_gpu(args, total, N, dev, a, utils_match, synthetic)
        except Exception as e:               # the line must survive (e.g. 8192 x 2048 pairs not fitting beside the shard)
            n1 = {"error": repr(e)}
    roofline = roofline_block(B, N, a.steps, iters_done, icp_ms, icp_launches, dt, _lib.BUILD_INFO, steps_timed)

    extras = {}
    if not a.no_extras and world == 1 and workload == "config2":   # N > 1 runs measure scaling, nothing else
        try:
            extras = extra_measurements(args, src, dst, T, dev, a)
        except Exception as e:               # the headline line must survive a failing side measurement
            extras = {"error": repr(e)}

    cpu = None
    if a.cpu_pairs > 0 and world == 1 and workload == "config2":
        try:
            cpu = cpu_baseline(S, D, a)
        except Exception as e:                   # e.g. no C compiler for the oracle on this box
            cpu = {"value": None, "unit": "registrations/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(e)}

    name = ("BASELINE config 2: synthetic %d cluster pairs x %d pts per GPU" % (B, N) if workload == "config2" else
            "BASELINE config 4: synthetic %d cluster pairs x %d pts in total, %s per GPU" % (total, N, "/".join(map(str, sorted(set(counts))))))
    out = {
        "metric": "cluster-pair ICP registrations/sec", "value": round(value, 2), "unit": "registrations/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{name}, <= {a.iters} ICP iters ({a.stop_mode} stop), thres_dist 0.1, "
                               f"translation_frame 2.0 (41x41x3 bins)" +
                               ("; NOTE: `--gpus 1` runs config 2 (the headline) and is NOT the N = 1 point of the config-4 curve that "
                                "`--gpus N > 1` runs -- that point is `config4_on_one_gpu` here and `n1_same_workload` on every N > 1 line"
                                if workload == "config2" and world == 1 and a.workload == "auto" else
                                "; strong scaling of ONE workload: its single-GPU point is this line's `n1_same_workload` (not the `--gpus 1` "
                                "line, which is config 2)" if workload == "config4" and world > 1 else ""),
                   "pairs_total": total, "pairs_per_gpu": counts, "points": N, "icp_iteration_cap": a.iters,
                   "stop_mode": a.stop_mode,
                   "sharding": (f"contiguous blocks of pairs over {world} rank(s); every step = hist_icp + match_eval + ONE "
                                f"RCCL all_gather of the [B,26] rows (transform + 40-byte pair row) inside the timed region")
                   if collective else "one rank, no exchange step"},
        "library_build": _lib.BUILD_INFO,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "extras": extras,
    }
    if world == 1 and workload == "config2" and not a.no_extras:
        try:
            out["config4_on_one_gpu"] = config4_single_gpu(dev, a)
        except Exception as e:
            out["config4_on_one_gpu"] = {"error": repr(e)}
    if gather_check is not None:
        out["gather_check"] = gather_check
    if collective and workload == "config4":
        out["n1_same_workload"] = n1
        out["scaling_efficiency"] = (round(value / (world * n1["value"]), 4) if n1 and n1.get("value") else None)
    print(json.dumps(out))
    if collective:
        dist.destroy_process_group()


def same_workload_on_one_gpu(args, total, N, dev, a, utils_match, synthetic, steps=5):
    """The config-4 step of `--gpus N` on ONE GPU: all `total` pairs as one batch through hist_icp_eval (registration + the
    40-byte pair rows; no all_gather: one rank has nobody to gather from), `steps` steps after one warm-up, synchronised on both
    sides.  -> {value (registrations/s), ms_per_step, steps, pairs, what}"""
    S, D, _ = synthetic.make_batch(total, N, seed=0)
    s, d = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    utils_match.hist_icp_eval(args, s, d, return_iterations=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        utils_match.hist_icp_eval(args, s, d, return_iterations=True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"value": round(total * steps / dt, 2), "unit": "registrations/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "pairs": total, "what": "the same step (hist_icp + match_eval over all pairs, one batch) on rank 0's GPU alone, after the "
                                    "timed region; scaling_efficiency = value / (n_gpus * n1_same_workload.value)"}


def roofline_block(B, N, steps, iters_done, icp_ms, icp_launches, dt, build, steps_timed=None):
    """Roofline of the dominant kernel, icp_kernel.  One launch runs ALL ICP iterations of the batch (speculative
    execution of the batch-global stop), or one launch per iteration beyond 128 iterations.
    Algorithmic work per executed iteration: P = (n_s + n_d) * 16 B and E = n_s * n_d pair-evaluations
    (8 lane-ops each) per pair.  Top-level achieved/peak/frac follow the bench contract (algorithmic HBM bytes /
    launch duration); `bound` names what actually binds the kernel and `valu` carries that roofline."""
    P, E = (N + N) * 16, N * N
    steps_all, steps = steps, (steps_timed or steps)    # the steps whose launches carried events (timed_steps: one in `every`)
    executed = iters_done * steps                       # iterations of the batch rule (same every step)
    alg_bytes = B * P * executed
    alg_evals = B * E * executed
    launches = max(icp_launches, 1)
    sec = icp_ms * 1e-3
    gbs = alg_bytes / sec / 1e9 if sec > 0 else 0.0
    lane_alg = alg_evals * LANE_OPS_PER_EVAL / sec if sec > 0 else 0.0
    counters, note = None, "no counters file"
    if os.path.exists(COUNTERS_FILE):
        try:
            cj = json.load(open(COUNTERS_FILE))
            shape = "all_iterations" if icp_launches == steps else "one_iteration"
            if cj.get("library_build") != build:
                note = f"{os.path.basename(COUNTERS_FILE)} was collected with library build {cj.get('library_build')}, this is {build}: refused"
            elif (cj.get("launch_shape"), cj.get("pairs"), cj.get("points")) != (shape, B, N):
                note = "counters file describes another launch shape: refused"
            else:
                counters, note = cj, "PMC passes of the same library build (profiles/, rocprofv3 --pmc, separate passes)"
        except Exception as e:
            note = "unreadable counters file: " + repr(e)
    valu = {"peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
            "algorithmic_pair_evaluations_per_s": round(alg_evals / sec, 1) if sec > 0 else 0.0,
            "algorithmic_lane_ops_per_s": round(lane_alg, 1),
            "algorithmic_frac": round(lane_alg / VALU_PEAK_LANE_OPS, 4),
            "algorithmic_note": "the brute-force formulation's n_s*n_d evaluations per iteration; the exact sorted sweep "
                                "executes ~12 % of them and the periodic fast-forward skips repeated iterations, so this "
                                "may exceed 1",
            "SQ_INSTS_VALU_per_launch": None, "executed_lane_ops_per_s": None, "executed_frac": None}
    if counters is not None and counters.get("SQ_INSTS_VALU_per_launch"):
        insts = float(counters["SQ_INSTS_VALU_per_launch"])
        lane_exec = insts * 64.0 / (sec / launches)
        valu.update({"SQ_INSTS_VALU_per_launch": insts, "executed_lane_ops_per_s": round(lane_exec, 1),
                     "executed_frac": round(lane_exec / VALU_PEAK_LANE_OPS, 4)})
    return {
        "bound": "valu",
        "bound_note": "what binds: the launch is paced by its slowest pairs (one workgroup each, 36-50 iterations): VALU "
                      "issue of the search phase (neighbour certificates, probe rounds, moment sums; four waves per SIMD; "
                      "window scans in the first iterations) and the serial rotation solve of one wave per pair -- not HBM "
                      "(both clouds stay in LDS for all iterations; "
                      "arithmetic intensity ~256 lane-op/B against a ridge of ~10).  achieved/peak/frac below are the "
                      "contract's HBM figures (algorithmic bytes / launch duration); see `valu` for the binding roofline",
        "kernel": "icp_kernel (all ICP iterations of the batch per launch)" if icp_launches == steps
        else "icp_kernel (one ICP iteration of the batch per launch)",
        "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6),
        "traffic": counters.get("hbm_bytes_per_launch") if counters else None,
        "traffic_source": note,
        "algorithmic_bytes_per_launch": int(alg_bytes / launches),
        "algorithmic_bytes_per_iteration": B * P,
        "avg_launch_ms": round(icp_ms / launches, 5), "launches_timed": icp_launches,
        "steps_with_timed_launches": steps, "steps_of_the_timed_region": steps_all,
        "icp_iterations_executed_per_step": iters_done,
        "icp_share_of_step": round((icp_ms / steps) / (dt * 1e3 / steps_all), 4),
        "valu": valu,
    }


def config4_single_gpu(dev, a):
    """The N = 1 point of the config-4 scaling curve (what `--gpus N` runs, on one GPU), and the per-GPU shard shape
    of the 8-GPU run (1024 pairs x 2048 points) on its own."""
    from types import SimpleNamespace
    from icp_flow_amd import synthetic, utils_match
    N = 2048
    args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N,
                           icp_max_iterations=a.iters, icp_stop_mode=a.stop_mode)
    S, D, _ = synthetic.make_batch(8192, N, seed=0)
    src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    out = {"workload": "BASELINE config 4 (8192 cluster pairs x 2048 pts, <= %d iters), same timing protocol" % a.iters}
    for tag, nb, steps in (("all_8192_pairs", 8192, 5), ("shard_of_8_gpus_1024_pairs", 1024, 10)):
        s, d = src[:nb].contiguous(), dst[:nb].contiguous()
        dt, icp_ms, launches, _, iters, _ = timed_steps(lambda: utils_match.hist_icp(args, s, d, return_iterations=True),
                                                     lambda: torch.cuda.synchronize(dev), steps, 1, a.iters)
        out[tag] = {"registrations_per_s": round(nb * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
                    "icp_iterations": int(iters.item()), "icp_kernel_ms_per_step": round(icp_ms / steps, 3)}

        def step_eval():   # the step `--gpus N` times on every rank, less the all_gather: hist_icp + match_eval (one call)
            T, _, it = utils_match.hist_icp_eval(args, s, d, return_iterations=True)
            return T, it
        dt_eval, _, _, _, _, _ = timed_steps(step_eval, lambda: torch.cuda.synchronize(dev), steps, 1, a.iters)
        out[tag]["registrations_per_s_with_match_eval"] = round(nb * steps / dt_eval, 1)
        if nb == 1024:
            out[tag]["roofline"] = config4_roofline(nb, N, int(iters.item()), dt / steps * 1e3, icp_ms / steps)
    return out


def config4_roofline(B, N, iters, step_ms, icp_ms):
    """Roofline of config 4's per-GPU shard, kernel by kernel.  Top level: the contract's figures for the
    dominant kernel (icp_kernel: algorithmic bytes I*B*P / its launch duration, live HIP events).  `kernels`: per kernel of
    the step the duration, the algorithmic HBM bytes (inputs once + outputs once), the HBM bytes the PMC passes count
    (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md) and the executed VALU fraction (SQ_INSTS_VALU x 64 / duration / 78.6 T
    lane-op/s) -- from profiles/r06_config4_shard_counters.json (tools/profile_workload.sh + summarize_workload.py), refused when
    it was collected with another build of the library."""
    from icp_flow_amd import _lib
    P = 2 * N * 16                                        # both clouds of a pair, 16 B a point
    L = 41 * 41 * 3 * 4                                   # a pair's vote bins, uint32
    alg = {"icp_kernel": ("all ICP iterations (I x B x P, SURVEY 8(d))", iters * B * P),
           "hist_vote_sorted_kernel": ("both z-sorted clouds in, the bins out", B * (P + L)),
           "sort_clouds_kernel": ("both clouds in, sorted float4 + SoA images out", B * (P + P + P * 3 // 4)),
           "zsort_kernel": ("both clouds in, z-sorted out", B * 2 * P),
           "sweep_scan_kernel<0": ("six of the twelve scoring scans: both SoA images in per scan", 6 * B * P * 3 // 4),
           "sweep_scan_kernel<1": ("roll-back check: two scans", 2 * B * P),
           "hist_peaks_kernel": ("the bins in, five peaks out", B * L)}
    gbs = iters * B * P / (icp_ms * 1e-3) / 1e9 if icp_ms > 0 else 0.0
    out = {"bound": "valu", "kernel": "icp_kernel<512, PERSIST, HELP> (all ICP iterations of the shard in one launch)",
           "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6),
           "algorithmic_bytes_per_launch": iters * B * P, "avg_launch_ms": round(icp_ms, 4), "icp_share_of_step": round(icp_ms / step_ms, 4),
           "traffic": None, "traffic_source": "no counters file", "kernels": None}
    if os.path.exists(CONFIG4_COUNTERS_FILE):
        try:
            cj = json.load(open(CONFIG4_COUNTERS_FILE))
            if cj.get("library_build") != _lib.BUILD_INFO:
                out["traffic_source"] = (f"{os.path.basename(CONFIG4_COUNTERS_FILE)} was collected with library build "
                                         f"{cj.get('library_build')}, this is {_lib.BUILD_INFO}: refused")
            else:
                out["traffic_source"] = "PMC passes of the same library build (profiles/, rocprofv3 --pmc, separate passes; read side x2)"
                ks = []
                for name, e in sorted(cj["kernels"].items(), key=lambda kv: -kv[1].get("us_per_step", 0.0)):
                    key = next((k for k in alg if name.startswith(k)), None)
                    us = e.get("avg_us")
                    row = {"kernel": name, "launches_per_step": e.get("launches_per_step"), "avg_us": round(us, 1) if us else None,
                           "us_per_step": round(e.get("us_per_step", 0.0), 1), "hbm_bytes_counters": e.get("hbm_bytes_per_dispatch"),
                           "executed_valu_frac": e.get("executed_valu_frac"), "wait_frac": e.get("wait_frac"),
                           "lds_bank_conflict_frac": e.get("lds_bank_conflict_frac")}
                    if key is not None and us:
                        row["algorithmic_bytes"] = alg[key][1]
                        row["algorithmic_bytes_are"] = alg[key][0]
                        row["hbm_frac_algorithmic"] = round(alg[key][1] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                    if e.get("hbm_bytes_per_dispatch") and us:
                        row["hbm_frac_counters"] = round(e["hbm_bytes_per_dispatch"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                    ks.append(row)
                    if name.startswith("icp_kernel"):
                        out["traffic"] = e.get("hbm_bytes_per_dispatch")
                        out["valu"] = {"peak_lane_ops_per_s": VALU_PEAK_LANE_OPS, "SQ_INSTS_VALU_per_launch": e.get("SQ_INSTS_VALU_per_dispatch"),
                                       "executed_frac": e.get("executed_valu_frac")}
                out["kernels"] = ks
                out["kernel_us_per_step_under_the_profiler"] = cj.get("kernel_us_per_step")
        except Exception as e:
            out["traffic_source"] = "unreadable counters file: " + repr(e)
    return out


def extra_measurements(args, src, dst, T, dev, a):
    """Untimed-region extras (not part of `value`): match_eval, per-pair stop mode, ICP only."""
    from types import SimpleNamespace
    from icp_flow_amd import utils_icp, utils_match

    def timeit(fn, reps=10):
        fn()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / reps * 1e3

    from icp_flow_amd import _lib
    B = src.shape[0]
    # the frame pair FIRST: its latency (one frame pair at a time, stage 2's initial poses on a second stream beside stage 1's ICP)
    # is the one measurement here that depends on which hardware queues the process's streams share (HIP multiplexes streams
    # onto four queues by default; measured: 1.40 / 1.75 ms in a process that has created no other streams -- or anywhere with
    # GPU_MAX_HW_QUEUES=16 -- against 1.6 / 1.95 ms after hist_icp_many's and the in-flight workers' streams exist), and a
    # frame-pair pipeline is a process of its own
    fp_first = frame_pair_measurement(dev)
    out = {"match_eval_ms_per_batch": round(timeit(lambda: utils_match.match_eval(args, src, dst, T)), 4)}
    # the same registration with the ICP loop's correspondence search forced to the all-pairs LDS scan
    # (the north star's brute-force formulation; identical results)
    with _lib.options(search="scan"):
        ms = timeit(lambda: utils_match.hist_icp(args, src, dst), reps=5)
    out["all_pairs_scan_search_registrations_per_s"] = round(B / ms * 1e3, 1)
    fast = SimpleNamespace(**{**vars(args), "icp_stop_mode": "per_pair"})
    ms = timeit(lambda: utils_match.hist_icp(fast, src, dst))
    out["per_pair_stop_registrations_per_s"] = round(B / ms * 1e3, 1)
    init = torch.eye(4, device=dev)[None].repeat(B, 1, 1).contiguous()
    ms = timeit(lambda: utils_icp.apply_icp(args, src, dst, init))
    out["icp_only_apply_icp_registrations_per_s"] = round(B / ms * 1e3, 1)
    # the two drop-in primitives on their own (the reference's native boundaries: hist_cuda.hist, knn_points)
    from icp_flow_amd import hist as hip_hist, utils_helper, utils_hist
    ex, ey, ez = utils_hist.bin_edges(args)
    ms = timeit(lambda: hip_hist.hist(dst, src, ex.min(), ey.min(), ez.min(), ex.max(), ey.max(), ez.max(), len(ex),
                                     len(ey), len(ez)))
    out["hist_all_pairs_vote_ms_per_batch"] = round(ms, 4)
    ms = timeit(lambda: utils_helper.nearest_neighbor_batch(src, dst))
    out["nearest_neighbor_batch_ms_per_batch"] = round(ms, 4)
    # capacity with several INDEPENDENT batches in flight through ONE call (icpflow_hist_icp_many / hist_icp_many): four
    # different 256-pair config-2 batches (pairs 0..1023 of the generator), each one registration with its own
    # batch-global stop rule and results identical to a call of its own; the tail of one batch's ICP launch -- few pairs
    # still iterating, most CUs idle -- overlaps the vote and scoring of the others.  Not the headline: `value` times
    # one batch after the other on one stream.
    from icp_flow_amd import synthetic
    many = [synthetic.make_batch(B, src.shape[1], seed=0, first=k * B) for k in range(4)]
    srcs = [torch.from_numpy(m[0]).to(dev) for m in many]
    dsts = [torch.from_numpy(m[1]).to(dev) for m in many]
    separate = [utils_match.hist_icp(args, a_, b_) for a_, b_ in zip(srcs, dsts)]
    ms, outs = None, None
    for reps in (2, 10):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(reps):
            outs = utils_match.hist_icp_many(args, srcs, dsts)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t) / reps * 1e3
    out["four_batches_in_one_call_registrations_per_s"] = round(4 * B / ms * 1e3, 1)
    out["four_batches_in_one_call_identical_to_separate_calls"] = bool(all(torch.equal(a_, b_) for a_, b_ in zip(outs, separate)))
    ms = timeit(lambda: [utils_match.hist_icp(args, a_, b_) for a_, b_ in zip(srcs, dsts)], reps=10)
    out["the_same_four_batches_one_after_the_other_registrations_per_s"] = round(4 * B / ms * 1e3, 1)
    try:
        out["ragged_real_shape"] = ragged_real_shape(dev)
    except Exception as e:
        out["ragged_real_shape"] = {"error": repr(e)}
    try:
        # the same distribution of cluster sizes with the two clouds of a pair alike (round 3: the independent draw pits
        # 35-point clusters against 7000-point ones, which no candidate pair of match_pcds is after sanity_check)
        out["ragged_real_shape_matched_sizes"] = ragged_real_shape(dev, sizes="matched")
    except Exception as e:
        out["ragged_real_shape_matched_sizes"] = {"error": repr(e)}
    if fp_first is not None:
        out["frame_pair"] = fp_first
    return out


def ragged_real_shape(dev, B=128, N=10000, cap=100, sizes=True):
    """SURVEY 8(d) "ragged variant": the shape the real sweeps (BASELINE configs 3 and 5) present -- clusters of
    n ~ logUniform(20, 10^4) points padded to max_points = 10000 (main.sh:10) with (1e8, 1e8, 1e8, 0), a frame's worth of
    candidate pairs per batch (B = 128), the reference's 100-iteration cap (utils_icp.py:54).  Registrations/s of
    hist_icp and a coarse split by entry point (each timed on its own): estimate_init_pose (vote, peaks, 12 scoring
    scans), apply_icp (ICP from those poses + roll-back check), match_eval; the ICP kernel's share by HIP events."""
    from types import SimpleNamespace
    from icp_flow_amd import _lib, synthetic, utils_hist, utils_icp, utils_match
    S, D, _ = synthetic.make_batch(B, N, seed=0, ragged=sizes, n_min=20)
    src, dst = torch.from_numpy(S).to(dev), torch.from_numpy(D).to(dev)
    args = SimpleNamespace(thres_dist=0.1, translation_frame=2.0, chunk_size=50, max_points=N, icp_max_iterations=cap,
                           icp_stop_mode="reference")

    def timeit(fn, reps=5):
        fn()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / reps * 1e3, r

    prof = _lib.Profile(64)
    with _lib.options(profile=prof):
        ms, (T, iters) = timeit(lambda: utils_match.hist_icp(args, src, dst, return_iterations=True))
    icp_ms, launches = prof.collect()
    prof.close()
    ms_init, init = timeit(lambda: utils_hist.estimate_init_pose(args, src, dst))
    ms_icp, _ = timeit(lambda: utils_icp.apply_icp(args, src, dst, init))
    ms_eval, _ = timeit(lambda: utils_match.match_eval(args, src, dst, T))
    n = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
    roof = ragged_roofline(n[0], n[1], int(iters.item()), icp_ms / 6, ms, "matched" if sizes == "matched" else "independent", _lib.BUILD_INFO)
    how = ("both clouds of a pair draw their sizes independently" if sizes is True else
           "n_dst = n_src * U(0.8, 1.25): the same object from two ranges, as candidate pairs are after sanity_check")
    return {"workload": f"{B} cluster pairs, n ~ logUniform(20, {N}) padded to {N} ({how}), <= {cap} ICP iterations (reference stop)",
            "points_per_cluster_median": int(np.median(np.concatenate(n))), "points_per_cluster_max": int(max(n[0].max(), n[1].max())),
            "registrations_per_s": round(B / ms * 1e3, 1), "ms_per_batch": round(ms, 3), "icp_iterations": int(iters.item()),
            "icp_kernel_ms_per_batch": round(icp_ms / 6, 3),   # (6 calls: one warm-up + 5 timed)
            "split_ms": {"estimate_init_pose": round(ms_init, 3), "apply_icp": round(ms_icp, 3), "match_eval": round(ms_eval, 3)},
            "roofline": roof}


def ragged_roofline(ns, nd, iters, icp_ms, batch_ms, sizes, build):
    """Roofline of the ragged real-shape batch, in SURVEY 8(d)'s terms with the VALID lengths of every
    pair (not the padded width): per scan P_b = (n_s + n_d) * 16 B and E_b = n_s * n_d pair-evaluations; the ICP launch
    runs I iterations of the batch rule, a registration 15 + I scans.  HBM contract figure = algorithmic bytes of the ICP
    launch / its duration (HIP events on its stream); the roofline that binds is VALU (the counters: SQ_INSTS_VALU x 64
    lanes against 256 CU x 128 lanes x 2.4 GHz), from profiles/r04_ragged_counters.json when it describes this build."""
    ns, nd = ns.astype(np.int64), nd.astype(np.int64)
    P, E = int(((ns + nd) * 16).sum()), int((ns * nd).sum())
    sec = icp_ms * 1e-3
    alg_bytes_icp = P * iters
    out = {"bound": "valu", "kernel": "icp_kernel<768, TEAM> (all ICP iterations of the batch in one launch; teams of workgroups on the long pairs)",
           "achieved": round(alg_bytes_icp / sec / 1e9, 3) if sec > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(alg_bytes_icp / sec / 1e9 / HBM_PEAK_GBS, 6) if sec > 0 else 0.0,
           "algorithmic_bytes_per_launch": alg_bytes_icp, "algorithmic_bytes_per_registration_batch": P * (15 + iters),
           "algorithmic_pair_evaluations_per_launch": E * iters, "sum_valid_points": int((ns + nd).sum()),
           "avg_launch_ms": round(icp_ms, 4), "icp_share_of_batch": round(icp_ms / batch_ms, 4) if batch_ms > 0 else None,
           "traffic": None, "traffic_source": "no counters file", "valu": None}
    if os.path.exists(RAGGED_COUNTERS_FILE):
        try:
            cj = json.load(open(RAGGED_COUNTERS_FILE))
            if cj.get("library_build") != build:
                out["traffic_source"] = (f"{os.path.basename(RAGGED_COUNTERS_FILE)} was collected with library build "
                                         f"{cj.get('library_build')}, this is {build}: refused")
            elif sizes in cj:
                k = next((v for name, v in cj[sizes]["kernels"].items() if name.startswith("icp_kernel")), None)
                pb = cj[sizes]["per_batch"]
                out["traffic_source"] = "PMC passes of the same library build (profiles/, rocprofv3 --pmc, separate passes; read side x2)"
                if k is not None:
                    out["traffic"] = k.get("hbm_bytes_per_dispatch")
                    insts = k.get("SQ_INSTS_VALU_per_dispatch")
                    if insts and sec > 0:
                        lane = insts * 64.0 / sec
                        out["valu"] = {"peak_lane_ops_per_s": VALU_PEAK_LANE_OPS, "SQ_INSTS_VALU_per_launch": insts,
                                       "executed_lane_ops_per_s": round(lane, 1), "executed_frac": round(lane / VALU_PEAK_LANE_OPS, 4),
                                       "algorithmic_lane_ops_per_s": round(E * iters * LANE_OPS_PER_EVAL / sec, 1),
                                       "algorithmic_frac": round(E * iters * LANE_OPS_PER_EVAL / sec / VALU_PEAK_LANE_OPS, 4)}
                if batch_ms > 0 and pb.get("SQ_INSTS_VALU"):
                    out["whole_batch"] = {"hbm_bytes_counters": pb.get("hbm_bytes"), "hbm_bytes_algorithmic": P * (15 + iters),
                                          "SQ_INSTS_VALU": pb["SQ_INSTS_VALU"],
                                          "executed_valu_frac": round(pb["SQ_INSTS_VALU"] * 64.0 / (batch_ms * 1e-3) / VALU_PEAK_LANE_OPS, 4),
                                          "hbm_frac_counters": round(pb.get("hbm_bytes", 0) / (batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        except Exception as e:
            out["traffic_source"] = "unreadable counters file: " + repr(e)
    return out


def frame_pair_measurement(dev):
    """BASELINE config 1: ms / frame pair on the reference's demo frame pair (tests/golden/g8_demo*.npz:
    63 k points per frame, labels of the fixture, both association stages + per-point flow; clouds
    resident, flags of demo.sh) and the agreement of the flow with the reference's own run."""
    gdir = os.path.join(REPO, "tests", "golden")
    try:
        g, lab = np.load(os.path.join(gdir, "g8_demo.npz")), np.load(os.path.join(gdir, "g8_demo_labels.npz"))
    except OSError:
        return None
    from icp_flow_amd import frame_pairs, utils_flow, utils_track
    G = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    ps, pd = G(g["point_src"]), G(g["point_dst"])
    ls, ld = G(lab["label_src"]).float(), G(lab["label_dst"]).float()
    ego = torch.eye(4, device=dev)                 # the ego pose: an input like the clouds (demo.py:217 passes the identity)
    res = {"data": "demo.npz frame pair of the reference, labels from the G8 fixture", "points": [len(ps), len(pd)]}
    kept = {}
    for mp in (int(g["max_points"]), 10000):
        a = frame_pairs.default_args(max_points=mp)

        # main.py:139 seeds torch's global generator once per run; a private generator with the same seed gives the same draws
        # (the random subsample of the over-long wall, utils_helper.py:198-201) -- torch.manual_seed() itself costs ~50 us a
        # call here (it walks every backend's lazy-init queue and formats a stack trace), which is not the product's time
        a.generator = torch.Generator()
        a.native_host = False          # (`run` below times the Python host; the native call is timed on its own)

        def run():
            a.generator.manual_seed(0)
            pairs, Tm = utils_track.track(a, ps, pd, ls, ld)
            return pairs, utils_flow.flow_estimation_torch(a, ps, pd, ls, ld, pairs, Tm, ego)

        run()
        # seven frame pairs timed one by one (a latency: each ends with a synchronisation), the median reported and every
        # run listed -- a mean over a handful of runs has been seen to carry one stall of the caching allocator (a
        # hipMalloc after the other extras: +2.5 ms on one run of five)
        runs = []
        for _ in range(7):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            pairs, flow = run()
            torch.cuda.synchronize(dev)
            runs.append(round((time.perf_counter() - t) * 1e3, 3))
        # the same through ONE call into the library (icpflow_track_frame: the host half of match_pcds in C++ as well; the
        # default of frame_pairs.register_frame_pair / register_in_flight): same bits
        def run_native():
            out = frame_pairs.track_frame_native(a, ps, pd, ls, ld, ego, ps, seed=0)
            return out["pairs"], out["flow"]

        run_native()
        runs_native = []
        for _ in range(7):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            pairs_native, flow_native = run_native()
            torch.cuda.synchronize(dev)
            runs_native.append(round((time.perf_counter() - t) * 1e3, 3))
        # the reference's own two calls, track() + flow_estimation_torch(), as a user of the drop-in makes them: track() goes
        # through the same native call (utils_match.match_pcds), the flow is a launch of its own
        a.native_host = True
        run()
        runs_api = []
        for _ in range(7):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            pairs_api, flow_api = run()
            torch.cuda.synchronize(dev)
            runs_api.append(round((time.perf_counter() - t) * 1e3, 3))
        a.native_host = False
        # (the host-side association of the Python host, for the record: same pairs, numbers to rounding)
        a.device_association = False
        pairs_host, flow_host = run()
        a.device_association = None
        entry = {"ms_per_frame_pair": sorted(runs_native)[len(runs_native) // 2], "ms_per_frame_pair_runs": runs_native,
                 "ms_per_frame_pair_python_host_runs": runs, "matched_cluster_pairs": int(len(pairs)),
                 "ms_per_frame_pair_python_host": sorted(runs)[len(runs) // 2],
                 "ms_per_frame_pair_track_then_flow": sorted(runs_api)[len(runs_api) // 2], "ms_per_frame_pair_track_then_flow_runs": runs_api,
                 "track_then_flow_identical_to_python_host": bool(torch.equal(pairs_api, pairs) and torch.equal(flow_api, flow)),
                 "association": "on the device (one read-back per frame pair); ms_per_frame_pair: one call into the library per frame pair (icpflow_track_frame), ms_per_frame_pair_track_then_flow: the drop-in's track() (same native call) + flow_estimation_torch(), ms_per_frame_pair_python_host: the same two with the Python host (args.native_host = False)",
                 "native_call_identical_to_python_host": bool(torch.equal(pairs_native, pairs) and torch.equal(flow_native, flow)),
                 "device_vs_host_association_same_pairs": bool(torch.equal(pairs[:, :2], pairs_host[:, :2])),
                 "device_vs_host_association_max_flow_difference_m": float((flow - flow_host).abs().max()),
                 "epe_vs_ground_truth_m": round(float(np.linalg.norm(flow.cpu().numpy() - g["gt_flow"], axis=1).mean()), 5)}
        # against the reference's own run at the same max_points: the G8 fixtures made with torch.topk's CUDA tie order
        # (tools/gen_golden.py topk_cuda_order -- the order of ATen's radix select, which is also the product's rule;
        # tests/test_gpu_parity.py::test_demo_frame_pair_track_and_flow_vs_reference) and, for the record, the fixtures
        # made with torch-CPU's order, under which the reference's own two runs differ by centimetres on the clusters
        # that are still moving when its stage-1 batch stops after 41 / 55 instead of 100 iterations.
        tag = "g8_demo" if mp == int(g["max_points"]) else ("g8_demo_mp10000" if mp == 10000 else None)
        if tag is not None:
            try:
                ref = np.load(os.path.join(gdir, tag + "_cudatopk.npz"))
                err = np.abs(flow.cpu().numpy() - ref["flow"]).max(axis=1)
                entry["max_flow_difference_to_reference_run_m"] = float(err.max())
                entry["points_within_1e-4_m_of_reference_run"] = int((err < 1e-4).sum())
                entry["reference_run"] = "reference's match_pcds + flow_estimation_torch, torch.topk in CUDA's tie order, stage iterations %s" % list(map(int, ref["stage_iterations"]))
                cpu = np.load(os.path.join(gdir, tag + ".npz"))
                entry["reference_runs_differ_between_tie_orders_by_m"] = float(np.abs(cpu["flow"] - ref["flow"]).max())
            except OSError:
                pass
        res[f"max_points_{mp}"] = entry
        kept[mp] = (a, flow, flow_host)
    # (the latencies of BOTH settings before any frame pair is put in flight: the workers' streams, once they exist, share the
    # process's few hardware queues with the streams of a single frame pair)
    for mp in (int(g["max_points"]), 10000):
        a, flow, flow_host = kept[mp]
        entry = res[f"max_points_{mp}"]
        # the same frame pair as a STREAM (BASELINE configs 3 / 5 are streams of independent frame pairs, main.py:184-215):
        # 12 copies with 4 in flight, 24 with 8 (frame_pairs.register_in_flight: one stream and one host thread per frame pair
        # in flight, each frame pair one blocking icpflow_track_frame call; "_python_scheduler": one host thread, generators,
        # asynchronous hand-overs, host-side association); wall time of the stream over its frame pairs, uploads of the
        # clouds included -- throughput, not the latency above
        fp_obj = frame_pairs.FramePair(g["point_src"], g["point_dst"], lab["label_src"], lab["label_dst"], None, g["gt_flow"])
        for in_flight in (4, 8):
            copies = [fp_obj] * (3 * in_flight)
            for native, suffix, want_flow in ((True, "", flow), (False, "_python_scheduler", flow_host)):
                a.native_host = native
                # (the untimed pass keeps every result for the comparison; the timed one consumes them as a sweep would --
                # holding twelve frames' outputs makes the caching allocator grow on every stream inside the timed region)
                flows = {i: o["flow"] for i, _, o in frame_pairs.register_in_flight(a, copies, dev, in_flight)}
                entry[f"stream_{in_flight}_in_flight_identical_flow{suffix}"] = bool(all(torch.equal(f, want_flow) for f in flows.values()))
                del flows
                # three timed passes of the stream, the median reported and every pass listed: a single pass has been seen
                # to catch a stall of the caching allocator (4.6 ms per frame pair once, after the other extras, against 1.4-1.8)
                passes = []
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t = time.perf_counter()
                    for _ in frame_pairs.register_in_flight(a, copies, dev, in_flight):
                        pass
                    torch.cuda.synchronize(dev)
                    passes.append(round((time.perf_counter() - t) / len(copies) * 1e3, 3))
                entry[f"stream_ms_per_frame_pair_{in_flight}_in_flight{suffix}"] = sorted(passes)[1]
                entry[f"stream_ms_per_frame_pair_{in_flight}_in_flight_passes{suffix}"] = passes
            a.native_host = True
        res[f"max_points_{mp}"] = entry
    res["cluster_dbscan"] = cluster_measurement(dev, g, gdir)
    res["cluster_hdbscan"] = hdbscan_measurement(dev, g, gdir)
    return res


def hdbscan_measurement(dev, g, gdir):
    """SURVEY 8(f) row 4, the branch the reference's scripts select (--if_hdbscan): cluster_pcd of the stacked demo
    frame pair; spanning tree on the GPU checked against the weights of sklearn's exact Prim tree (G11), labels
    against the reference run's (G8 label fixture: sklearn's HDBSCAN through the reference's cluster_pcd, whose
    wall time in the build container is recorded in G11 as the CPU figure)."""
    from types import SimpleNamespace
    from icp_flow_amd import utils_cluster
    try:
        g11, lab8 = np.load(os.path.join(gdir, "g11_hdbscan.npz")), np.load(os.path.join(gdir, "g8_demo_labels.npz"))
    except OSError:
        return None
    pts = torch.from_numpy(np.concatenate([g["point_dst"], g["point_src"]], axis=0)).to(dev)
    a = SimpleNamespace(min_cluster_size=20, num_clusters=200, if_hdbscan=True, epsilon=0.25)
    nonground = torch.ones(len(pts), dtype=torch.bool, device=dev)

    def timeit(fn, reps=5):
        fn()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / reps * 1e3, out

    ms_tree, tree = timeit(lambda: utils_cluster.hdbscan_mst(pts, 21))   # min_samples 20 of the hdbscan package: the point itself not counted
    ms_all, lab = timeit(lambda: utils_cluster.cluster_pcd(a, pts, nonground))
    want = np.concatenate([lab8["label_dst"], lab8["label_src"]]).astype(np.int64)
    got = lab.cpu().numpy().astype(np.int64)
    noise_both = int(((got < 0) & (want < 0)).sum())
    moved = int(((got < 0) != (want < 0)).sum())
    return {"points": int(len(pts)), "min_cluster_size": 20, "clusters": int(got.max() + 1),
            "ms_per_frame_pair": round(ms_all, 2), "ms_spanning_tree_gpu": round(ms_tree, 2),
            "tree_weights_equal_sklearn_prim_g11": bool(np.array_equal(np.sort(np.sqrt(tree["w2"].cpu().numpy())),
                                                                       g11["demo_tree_weights"])),
            "reference_run_clusters": int(want.max() + 1), "noise_points_both": noise_both,
            "points_noise_in_one_only": moved,
            "cpu_reference_run_s": round(float(g11["demo_cpu_seconds"]), 1),
            "cpu_reference_run": "sklearn HDBSCAN (exact Prim) through the reference's cluster_hdbscan, build container, 1 core"}


def cluster_measurement(dev, g, gdir):
    """cluster_pcd (DBSCAN branch, utils_cluster.py:32-63) of the stacked demo frame pair
    (demo.py:210), points resident, labels checked against G10 (the reference's own cluster_pcd run with the
    open3d stand-in) and the CPU restatement (oracle/cluster.py, one thread) timed beside it."""
    from types import SimpleNamespace
    from icp_flow_amd import utils_cluster
    try:
        g10 = np.load(os.path.join(gdir, "g10_dbscan.npz"))
    except OSError:
        return None
    pts_h = np.concatenate([g["point_dst"], g["point_src"]], axis=0)
    pts = torch.from_numpy(pts_h).to(dev)
    eps, mcs, ncl = g10["demo_a_params"]
    a = SimpleNamespace(epsilon=float(eps), min_cluster_size=int(mcs), num_clusters=int(ncl), if_hdbscan=False)
    nonground = torch.ones(len(pts), dtype=torch.bool, device=dev)
    lab = utils_cluster.cluster_pcd(a, pts, nonground)
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(10):
        lab = utils_cluster.cluster_pcd(a, pts, nonground)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t) / 10 * 1e3
    t = time.perf_counter()
    for _ in range(10):
        utils_cluster.dbscan(pts, a.epsilon, a.min_cluster_size)
    torch.cuda.synchronize(dev)
    ms_kernels = (time.perf_counter() - t) / 10 * 1e3
    from oracle import cluster as oc
    t = time.perf_counter()
    want = oc.cluster_pcd(a, pts_h, np.ones(len(pts_h), dtype=bool))
    cpu_ms = (time.perf_counter() - t) * 1e3
    got = lab.cpu().numpy()
    return {"points": int(len(pts)), "eps": float(eps), "min_points": int(mcs), "clusters_kept": int(len(np.unique(got[got >= 0]))),
            "ms_per_frame_pair": round(ms, 3), "ms_icpflow_dbscan_only": round(ms_kernels, 3),
            "labels_equal_reference_run_g10": bool(np.array_equal(got.astype(np.int32), g10["demo_a_labels"])),
            "labels_equal_cpu_port": bool(np.array_equal(got, want)),
            "cpu_port_ms": round(cpu_ms, 1), "cpu_port": "oracle/cluster.py (scipy cKDTree + connected_components), 1 thread"}


def cpu_baseline(S, D, a):
    """The oracle (CPU port of the reference's algorithm: padded [B,N,4] clouds, 1 vote + 12 scoring scans + I ICP
    iterations + 2 roll-back scans) on a bounded sample of the SAME batch.  The port's inner loops (vote, K=1 NN) are
    C with OpenMP over pairs (oracle_core.c), the rest torch-CPU ops -- a FASTER baseline than the reference's own
    Python on CPU tensors (0.115 registrations/s on 8 cores at max_points 10000).  Timed at 32 threads
    (where the many small torch ops stop scaling) and with os.cpu_count() threads in the C loops; `value` is the
    better of the two."""
    from oracle import core as ocore
    from oracle import reference_path as rp
    n = min(a.cpu_pairs, S.shape[0])
    s, d = torch.from_numpy(S[:n]), torch.from_numpy(D[:n])
    args = rp.default_args(max_points=S.shape[1])
    runs = []
    # (OpenMP threads of the C loops, torch intra-op threads).  All 256 cores for BOTH was measured: 0.98
    # registrations/s -- the oracle's many small torch ops drown in fork/join -- so the second run gives every core
    # to the C loops (they parallelise over pairs) and keeps torch at 32.
    ncore = os.cpu_count() or 1
    for omp, tth in sorted({(min(ncore, 32), min(ncore, 32)), (ncore, min(ncore, 32))}):
        torch.set_num_threads(tth)
        ocore.set_num_threads(omp)
        rp.hist_icp(args, s[:2], d[:2], max_iterations=a.iters)              # warm-up (page-in, threads)
        t = time.perf_counter()
        _, aux = rp.hist_icp(args, s, d, max_iterations=a.iters, return_aux=True)
        dt = time.perf_counter() - t
        runs.append({"threads": omp, "torch_threads": tth, "registrations_per_s": round(n / dt, 3), "wall_s": round(dt, 2),
                     "omp_threads": ocore.num_threads(), "icp_iterations": aux["iterations"]})
    best = max(runs, key=lambda r: r["registrations_per_s"])
    return {"value": best["registrations_per_s"], "unit": "registrations/s", "cores": best["threads"],
            "host_cores_available": os.cpu_count(), "kind": "port",
            "sample": f"first {n} of the {S.shape[0]} pairs of the same batch, same {a.iters}-iteration cap "
                      f"(batch-global stop inside the sample after {best['icp_iterations']} iterations), "
                      f"{best['wall_s']:.2f} s wall at {best['threads']} threads",
            "runs": runs}


if __name__ == "__main__":
    main()
