"""accuracy metrics (G9) and the frame-pair stream (format, sharding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from icp_flow_amd import frame_pairs, synthetic, utils_eval


def test_compute_epe_matches_reference_golden():
    g8, g9 = load_golden("g8_demo"), load_golden("g9_epe")
    flow, gt = g8["flow"], g8["gt_flow"]
    assert np.allclose(utils_eval.compute_epe_test(flow, gt), g9["whole"], rtol=0, atol=1e-12)
    assert np.allclose(utils_eval.compute_epe_test(flow, gt, g9["mask"]), g9["masked"], rtol=0, atol=1e-12)
    assert np.allclose(utils_eval.compute_epe_test(g9["tiny_pred"], g9["tiny_gt"]), g9["tiny"], rtol=0, atol=1e-12)
    h = int(g9["split"])
    meter = utils_eval.AverageMeter()
    per = []
    for sl in (slice(0, h), slice(h, len(flow))):
        m = utils_eval.compute_epe_test(flow[sl], gt[sl])
        per.append(m)
        meter.update(*m, sl.stop - sl.start)
    assert np.allclose(np.array(per, dtype=np.float64), g9["per_frame"], rtol=0, atol=1e-12)
    assert np.allclose(list(meter.averages().values()), g9["meter_avg"], rtol=0, atol=1e-12)
    assert np.isclose(utils_eval.average_meter([p[0] for p in per], [h, len(flow) - h]), float(g9["average_meter_epe"]),
                      rtol=0, atol=1e-12)
    assert meter.num == len(flow) and meter.num_data == [h, len(flow) - h]


def _tiny_stream(tmp_path, n=3):
    paths = []
    for k in range(n):
        d = synthetic.make_frame_pair(seed=k, n_objects=4, n_max=200, n_background=200)
        fp = frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"],
                                   d["gt_flow"])
        path = os.path.join(tmp_path, f"fp_{k:03d}.npz")
        frame_pairs.save_frame_pair(path, fp)
        paths.append(path)
    return paths


def test_frame_pair_formats_round_trip(tmp_path):
    d = synthetic.make_frame_pair(seed=3, n_objects=3, n_max=100, n_background=100)
    fp = frame_pairs.FramePair(d["points_src"], d["points_dst"], d["labels_src"], d["labels_dst"], d["pose"],
                               d["gt_flow"])
    p = os.path.join(tmp_path, "a.npz")
    frame_pairs.save_frame_pair(p, fp)
    back = frame_pairs.load_frame_pair(p)
    for k in ("points_src", "points_dst", "labels_src", "labels_dst", "pose", "gt_flow"):
        assert np.array_equal(getattr(back, k), getattr(fp, k)), k
    # the reference's Argoverse / demo keys (dataset_argo.py:34-53): valid-index selection applies
    ns, nd = len(fp.points_src), len(fp.points_dst)
    v0 = np.arange(0, ns, 2)
    v1 = np.arange(1, nd, 3)
    q = os.path.join(tmp_path, "b.npz")
    np.savez(q, pc1=fp.points_src, pc2=fp.points_dst, pc1_flows_valid_idx=v0, pc2_flows_valid_idx=v1,
             gt_flow_0_1=fp.gt_flow, labels_src=fp.labels_src[v0], labels_dst=fp.labels_dst[v1])
    argo = frame_pairs.load_frame_pair(q)
    assert np.array_equal(argo.points_src, fp.points_src[v0]) and np.array_equal(argo.points_dst, fp.points_dst[v1])
    assert np.array_equal(argo.gt_flow, fp.gt_flow[v0]) and np.array_equal(argo.pose, np.eye(4, dtype=np.float32))
    # labels must be one per point; a pair without labels loads (it is clustered on the GPU when the
    # arguments name a clustering, else registering it is an error) and keeps its non-ground flags
    ng = np.arange(ns) % 3 != 0
    np.savez(os.path.join(tmp_path, "c.npz"), points_src=fp.points_src, points_dst=fp.points_dst, nonground_src=ng)
    bare = frame_pairs.load_frame_pair(os.path.join(tmp_path, "c.npz"))
    assert bare.labels_src is None and bare.labels_dst is None and bare.nonground_dst is None
    assert np.array_equal(bare.nonground_src, ng)
    with pytest.raises(ValueError):
        frame_pairs.cluster_frame_pair(frame_pairs.default_args(), None, None)
    with pytest.raises(ValueError):
        frame_pairs.FramePair(fp.points_src, fp.points_dst, fp.labels_src[:-1], fp.labels_dst)
    with pytest.raises(ValueError):
        frame_pairs.FramePair(fp.points_src, fp.points_dst, fp.labels_src, None)
    with pytest.raises(ValueError):
        frame_pairs.FramePair(fp.points_src, fp.points_dst, nonground_src=ng[:-1])
    assert frame_pairs.list_frame_pairs(str(tmp_path)) == sorted(os.path.join(tmp_path, f) for f in ("a.npz", "b.npz", "c.npz"))


def test_round_robin_and_translation_frame():
    items = list(range(7))
    parts = [frame_pairs.shard_round_robin(items, r, 3) for r in range(3)]
    assert parts == [[0, 3, 6], [1, 4], [2, 5]]
    assert sorted(sum(parts, [])) == items
    with pytest.raises(ValueError):
        frame_pairs.shard_round_robin(items, 3, 3)
    pose = np.eye(4)
    pose[0:3, 3] = (3.0, 4.0, 0.0)
    a = frame_pairs.default_args(translation_frame=2.0)
    assert frame_pairs.frame_translation(a, pose) == 2.0                   # fixed (demo.py:205)
    a.speed = 1.67
    assert frame_pairs.frame_translation(a, pose) == 10.0                  # ego motion dominates (main.py:200)
    assert frame_pairs.frame_translation(a, np.eye(4), gap=2) == pytest.approx(6.68)


def _oracle_register(args, fp, device):
    from oracle import reference_path as rp
    G = torch.from_numpy
    torch.manual_seed(0)
    pairs, T = rp.match_pcds(args, G(fp.points_src), G(fp.points_dst), G(fp.labels_src), G(fp.labels_dst))
    flow = rp.flow_estimation_torch(G(fp.points_src), G(fp.labels_src), pairs, T, G(fp.pose))
    return dict(pairs=pairs, transformations=T, flow=flow)


def _stream_worker(rank, world, port, paths, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args = frame_pairs.default_args(max_points=256)
    s = frame_pairs.run_stream(args, paths, "cpu", rank, world, register_fn=_oracle_register)
    if rank == 0:
        np.save(out, np.array([s[k] for k in ("frame_pairs", "matched_cluster_pairs", "evaluated_points", "epe", "accs",
                                              "accr", "outlier", "Routlier")], dtype=np.float64))
    dist.destroy_process_group()


def test_stream_two_ranks_equals_single_process(tmp_path):
    """Round-robin sharding + the one all_reduce of the summary (gloo, world 2) reproduce the
    single-process accuracy summary; registration itself is the oracle here (no GPU)."""
    paths = _tiny_stream(str(tmp_path), 3)
    torch.set_num_threads(2)
    one = frame_pairs.run_stream(frame_pairs.default_args(max_points=256), paths, "cpu", register_fn=_oracle_register)
    assert one["frame_pairs"] == 3 and one["matched_cluster_pairs"] >= 9 and one["epe"] < 0.02
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = os.path.join(tmp_path, "two.npy")
    mp.spawn(_stream_worker, args=(2, port, paths, out), nprocs=2, join=True)
    two = np.load(out)
    want = np.array([one[k] for k in ("frame_pairs", "matched_cluster_pairs", "evaluated_points", "epe", "accs", "accr",
                                      "outlier", "Routlier")], dtype=np.float64)
    assert np.allclose(two, want, rtol=1e-12, atol=0)


def test_stream_eight_ranks_uneven_equals_single_process(tmp_path):
    """World size 8, 13 frame pairs (shares of 2 and 1): the round-robin shards and the one all_reduce
    reproduce the single-process summary; the registration is the oracle (no GPU)."""
    paths = _tiny_stream(str(tmp_path), 13)
    torch.set_num_threads(2)
    one = frame_pairs.run_stream(frame_pairs.default_args(max_points=256), paths, "cpu", register_fn=_oracle_register)
    assert one["frame_pairs"] == 13
    assert [len(frame_pairs.shard_round_robin(paths, r, 8)) for r in range(8)] == [2, 2, 2, 2, 2, 1, 1, 1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = os.path.join(tmp_path, "eight.npy")
    mp.spawn(_stream_worker, args=(8, port, paths, out), nprocs=8, join=True)
    eight = np.load(out)
    want = np.array([one[k] for k in ("frame_pairs", "matched_cluster_pairs", "evaluated_points", "epe", "accs", "accr",
                                      "outlier", "Routlier")], dtype=np.float64)
    assert np.array_equal(eight[:3], want[:3])                       # frame pairs, matched cluster pairs, evaluated points
    assert np.allclose(eight[3:], want[3:], rtol=1e-6, atol=0)       # (weighted means: the sums are added in another order)


def test_sequence_files_in_the_reference_waymo_format(tmp_path):
    """dataset_pca.py:41-45 keys -> num_frames - 1 frame pairs (frame j -> frame 0): ego compensation
    (utils_helper.py:89-93), range crop (dataset_pca.py:61-64), per-gap translation frame (main.py:200), the estimated
    poses of the reference's <split>_pose side file (dataset_pca.py:118-125) taking precedence, ground truth per gap."""
    d = synthetic.make_sequence(seed=5, num_frames=3, n_objects=4, n_max=120, n_background=150)
    os.makedirs(os.path.join(tmp_path, "val"))
    path = os.path.join(tmp_path, "val", "s0.npz")
    np.savez(path, **d, sd_labels=np.zeros(len(d["raw_points"])), fb_labels=np.zeros(len(d["raw_points"])))
    assert frame_pairs.is_sequence(path)
    a = frame_pairs.default_args(speed=0.8333, range_x=30.0, range_y=30.0)
    # the sample only carries GROUND-TRUTH ego poses: "auto" falls back to them with a warning and says so in the result;
    # asked for explicitly there is no warning; a missing estimated-pose file is an error when it is asked for
    with pytest.warns(UserWarning, match="GROUND-TRUTH"):
        fps = frame_pairs.load_any(path, a)
    assert all(fp.pose_source == "ego_motion_gt" for fp in fps)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert len(frame_pairs.load_any(path, a, pose_source="ego_motion_gt")) == 2
    with pytest.raises(FileNotFoundError):
        frame_pairs.load_any(path, a, pose_source="pose_file")
    with pytest.raises(KeyError):
        frame_pairs.load_any(path, a, pose_source="ego_motion")
    assert [fp.gap for fp in fps] == [1, 2] and all(fp.labels_src is None for fp in fps)
    raw, t, P = d["raw_points"].astype(np.float64), d["time_indice"], d["ego_motion_gt"]
    keep = (np.abs(raw[:, 0]) < 30.0) & (np.abs(raw[:, 1]) < 30.0)
    for fp in fps:
        m = keep & (t == fp.gap)
        hom = np.concatenate([raw[m], np.ones((int(m.sum()), 1))], axis=1)
        assert np.array_equal(fp.points_src, (hom @ P[fp.gap].T)[:, 0:3].astype(np.float32))
        assert np.array_equal(fp.points_src_raw, d["raw_points"][m]) and np.array_equal(fp.points_dst, d["raw_points"][keep & (t == 0)])
        assert np.array_equal(fp.gt_flow, d["scene_flow"][m]) and np.array_equal(fp.nonground_src, d["nonground"][m])
        tf = frame_pairs.frame_translation(a, fp.pose_exact, fp.gap)
        assert tf == max(0.8333 * fp.gap, np.linalg.norm(P[fp.gap][0:3, -1])) * 2                  # main.py:200, verbatim
    # side file with estimated poses
    os.makedirs(os.path.join(tmp_path, "val_pose"))
    est = P.copy()
    est[:, 0, 3] += 0.25
    np.savez(os.path.join(tmp_path, "val_pose", "s0.npz"), ego_motion=est)
    fps2 = frame_pairs.load_any(path, a)
    assert np.allclose(fps2[0].pose, est[1].astype(np.float32)) and not np.allclose(fps2[0].points_src, fps[0].points_src)
    assert all(fp.pose_source == "pose_file" for fp in fps2)
    # only a DIRECTORY named like a split is one: '<tmp>/latest/val/...' must look in '<tmp>/latest/val_pose', and a
    # 'val' inside another name ('interval') or inside the file name is not a split
    deep = os.path.join(tmp_path, "interval_test_data", "latest", "val")
    os.makedirs(deep)
    p2 = os.path.join(deep, "val_s0.npz")
    np.savez(p2, **d)
    assert frame_pairs._pose_file(p2) is None
    os.makedirs(os.path.join(tmp_path, "interval_test_data", "latest", "val_pose"))
    np.savez(os.path.join(tmp_path, "interval_test_data", "latest", "val_pose", "val_s0.npz"), ego_motion=est)
    assert frame_pairs._pose_file(p2) == os.path.join(tmp_path, "interval_test_data", "latest", "val_pose", "val_s0.npz")
    # a frame-pair file is still one pair
    one = os.path.join(tmp_path, "pair.npz")
    frame_pairs.save_frame_pair(one, frame_pairs.FramePair(fps[0].points_src, fps[0].points_dst))
    assert not frame_pairs.is_sequence(one) and len(frame_pairs.load_any(one)) == 1


def test_setdiff1d_equals_the_reference_statement():
    """utils_match.setdiff1d (utils_helper.py:172-183: labels of t1 not in t2, t2 a subset of t1, sorted) on numpy arrays
    -- the form match_pcds uses between its two stages -- and on tensors, against the oracle's restatement."""
    from icp_flow_amd import utils_match
    from oracle import reference_path as rp
    rng = np.random.default_rng(4)
    for trial in range(50):
        n = int(rng.integers(1, 200))
        t1 = rng.choice(np.arange(-3, 400), size=n, replace=False).astype(np.int64)
        if trial % 3 == 0:
            t1 = np.concatenate([t1, t1[: n // 2]])                      # repeated labels in t1
        t2 = rng.choice(t1, size=int(rng.integers(0, len(t1) + 1)), replace=False) if trial % 7 else t1[:0]
        want = rp.setdiff1d(torch.from_numpy(t1), torch.from_numpy(t2)).numpy()
        got = utils_match.setdiff1d(t1, t2)
        assert got.dtype == t1.dtype and np.array_equal(got, want), trial
        assert torch.equal(utils_match.setdiff1d(torch.from_numpy(t1), torch.from_numpy(t2)), torch.from_numpy(want))


def test_flat_result_buffer_of_a_stage_maps_to_the_arrays():
    """The one buffer a stage of match_pcds brings to the host (utils_match._hist_icp_eval_flat: transforms, errors, inliers,
    ratios, ious [B,2], translations, rotations [B,3], one int32 iteration count) and its views, host and tensor form."""
    from icp_flow_amd import utils_match
    B = 7
    rng = np.random.default_rng(1)
    parts = [rng.normal(size=(B, 4, 4)), *[rng.normal(size=(B, 2)) for _ in range(4)], *[rng.normal(size=(B, 3)) for _ in range(2)]]
    flat = np.concatenate([p.astype(np.float32).reshape(-1) for p in parts] + [np.array([37], np.int32).view(np.float32)])
    assert flat.shape == (30 * B + 1,)
    for r in (flat, torch.from_numpy(flat)):
        T, ev, iters = utils_match._eval_views(r, B)
        got = [T, *ev]
        assert [tuple(g.shape) for g in got] == [p.shape for p in parts]
        for g, p in zip(got, parts):
            assert np.array_equal(np.asarray(g), p.astype(np.float32))
        assert int(iters[0]) == 37 and len(iters) == 1


def test_ragged_batches_independent_and_matched_sizes():
    """synthetic.make_batch: the ragged generator of bench.py's real-shape lines -- independent sizes (pinned: the bench
    line of rounds 2 and 3 must stay the same batch) and matched sizes (n_dst within 0.8 ... 1.25 of n_src)."""
    S, D, _ = synthetic.make_batch(16, 10000, seed=0, ragged=True, n_min=20)
    ns, nd = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
    assert list(ns[:6]) == [478, 2462, 37, 56, 3324, 3698] and list(nd[:6]) == [147, 1177, 1875, 153, 51, 6619]
    S, D, _ = synthetic.make_batch(16, 10000, seed=0, ragged="matched", n_min=20)
    ms, md = (S[:, :, 3] > 0).sum(1), (D[:, :, 3] > 0).sum(1)
    assert np.array_equal(ms, ns)
    ratio = md / ms
    assert (md >= 20).all() and (md <= 10000).all() and ((ratio > 0.79) & (ratio < 1.26) | (md == 20)).all()
    assert (S[:, :, 3][np.arange(16), ms - 1] == 1).all() and (S[0, ms[0]:, 0] == 1e8).all()
