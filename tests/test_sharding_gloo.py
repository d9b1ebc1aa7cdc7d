"""N > 1 path on CPU: two gloo ranks shard a batch, compute with an injected function (the
oracle's estimate_init_pose -- no batch coupling, so shard results must equal the unsharded
run bit for bit) and all_gather the transforms in rank order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from icp_flow_amd import sharding, synthetic


def test_shard_range_partitions_everything():
    for total in (0, 1, 7, 8, 256, 8192, 8193):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_range(r, world, total) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in got) - min(c for _, c in got) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import reference_path as rp
    torch.set_num_threads(1)
    S, D, _ = synthetic.make_batch(total, 128, seed=5)
    args = rp.default_args(max_points=128)
    T = sharding.register_sharded(args, torch.from_numpy(S), torch.from_numpy(D), rank, world,
                                  rp.estimate_init_pose)
    if rank == 0:
        np.save(out_path, T.numpy())
    dist.destroy_process_group()


def _run(world, total, tmp_path):
    out = str(tmp_path / f"gathered_{world}_{total}.npy")
    mp.spawn(_worker, args=(world, _free_port(), total, out), nprocs=world, join=True)
    return np.load(out)


def test_two_rank_gather_equals_single_process(tmp_path):
    from oracle import reference_path as rp
    for total in (6, 5):                      # even and uneven shards
        got = _run(2, total, tmp_path)
        S, D, _ = synthetic.make_batch(total, 128, seed=5)
        want = rp.estimate_init_pose(rp.default_args(max_points=128), torch.from_numpy(S), torch.from_numpy(D))
        assert got.shape == (total, 4, 4)
        assert np.array_equal(got, want.numpy())


def _rows_worker(rank, world, port, total, out_path, force):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = sharding.shard_range(rank, world, total)
    g = torch.Generator().manual_seed(1234)
    T = torch.randn(total, 4, 4, generator=g)[first:first + count]
    ev = [torch.randn(total, 2, generator=g)[first:first + count] for _ in range(4)]
    rows = sharding.pack_rows(T, first, *ev)
    counts = [sharding.shard_range(r, world, total)[1] for r in range(world)]
    got = sharding.gather_results(rows, world, counts=counts, force_collective=force)
    if rank == 0:
        np.save(out_path, got.numpy())
    dist.destroy_process_group()


def test_packed_rows_gather_in_rank_order(tmp_path):
    """[B,26] rows (transform + 40-byte pair row) through the one collective of the path: two ranks with
    even and uneven shards, and a world of ONE rank with the collective forced (what the GPU test runs over RCCL)."""
    for world, total, force in ((2, 6, False), (2, 5, False), (1, 4, True)):
        out = str(tmp_path / f"rows_{world}_{total}.npy")
        mp.spawn(_rows_worker, args=(world, _free_port(), total, out, force), nprocs=world, join=True)
        got = np.load(out)
        g = torch.Generator().manual_seed(1234)
        T = torch.randn(total, 4, 4, generator=g)
        ev = [torch.randn(total, 2, generator=g) for _ in range(4)]
        want = sharding.pack_rows(T, 0, *ev)
        assert got.shape == (total, sharding.ROW_FLOATS)
        assert np.array_equal(got, want.numpy())
        Tg, rows = sharding.unpack_rows(torch.from_numpy(got))
        assert torch.equal(Tg, T) and torch.equal(rows[:, 0], torch.arange(total, dtype=torch.float32))
