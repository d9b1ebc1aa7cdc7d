"""Multi-GPU sharding of the registration path.

Cluster pairs (and frame pairs) are independent units: one process per GPU registers a
contiguous block of pairs and the only exchange step is one all_gather of the [B,4,4]
transforms (64 B per pair) at the end -- `torch.distributed` backend "nccl" is RCCL over xGMI
on ROCm; "gloo" works on CPU tensors and is what the CPU tests use.  There is no reduction on
the path, so no collective other than this gather exists.

Note on exact parity: the reference's ICP stops when EVERY pair of the batch satisfies the
relative-rmse test (utils_icp_pytorch3d.py:209), so the iteration count -- and through it the
transform of pairs that are still moving -- depends on which pairs share a batch.  Sharding a
batch therefore reproduces the reference run on each shard, not on the union.
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, total):
    """Contiguous block of `total` units owned by `rank`: -> (first, count).
    The first total % world ranks get one extra unit."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(total), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_results(local, world, group=None, counts=None):
    """all_gather per-pair result rows ([B_local, ...]) in rank order -> [sum B_local, ...].
    `counts` (rows of every rank, known to all ranks -- e.g. from shard_range) makes this ONE collective with
    no host synchronisation; without it the counts are exchanged first (one more collective and a device ->
    host read per call).  Uneven shards are padded to the largest one."""
    if world == 1:
        return local
    local = local.contiguous()
    if counts is None:
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n, group=group)
        counts = [int(c.item()) for c in got]
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[dist.get_rank(group)] != local.shape[0]:
        raise ValueError(f"gather_results: counts {counts} do not describe this rank's {local.shape[0]} rows")
    nmax = max(counts)
    if all(c == nmax for c in counts):
        out = local.new_empty((world * nmax,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    padded = local.new_zeros((nmax,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def register_sharded(args, src, dst, rank, world, register_fn, group=None):
    """Register this rank's block of `src`/`dst` ([B_total,N,4], already on the rank's device)
    with `register_fn(args, src_block, dst_block) -> [B_local,4,4]` and gather all transforms."""
    first, count = shard_range(rank, world, src.shape[0])
    T = register_fn(args, src[first:first + count], dst[first:first + count])
    counts = [shard_range(r, world, src.shape[0])[1] for r in range(world)]   # every rank can compute them
    return gather_results(T, world, group=group, counts=counts)
