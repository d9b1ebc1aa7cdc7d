"""Multi-GPU sharding of the registration path.

Cluster pairs (and frame pairs) are independent units: one process per GPU registers a
contiguous block of pairs and the only exchange step is one all_gather of the per-pair result
rows at the end -- the [4,4] transform (64 B) and the 10-column pair row of match_pairs
(utils_match.py:120-131: labels, errors, inliers, ratios, ious; 40 B), packed by `pack_rows`
into ONE [B,26] float32 tensor -- `torch.distributed` backend "nccl" is RCCL over xGMI
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


ROW_FLOATS = 26   # 16 (transform, row-major) + 10 (pair row)


def pack_rows(T, first, errors, inliers, ratios, ious, labels=None):
    """One [B,26] float32 row per registered pair: the [4,4] transform and the pair row of match_pairs
    (utils_match.py:120-131): (src label, dst label, errors 2, inliers 2, ratios 2, ious 2).  Synthetic batches have no
    cluster labels: the global pair index `first + b` stands for both (exact in float32 below 2^24)."""
    B = T.shape[0]
    if labels is None:
        k = torch.arange(first, first + B, device=T.device, dtype=torch.float32)[:, None]
        labels = torch.cat([k, k], dim=1)
    return torch.cat([T.reshape(B, 16), labels.to(T.dtype), errors, inliers, ratios, ious], dim=1).contiguous()


def unpack_rows(rows):
    """-> (transforms [B,4,4], pair rows [B,10])."""
    return rows[:, :16].reshape(-1, 4, 4), rows[:, 16:]


def gather_results(local, world, group=None, counts=None, force_collective=False):
    """all_gather per-pair result rows ([B_local, ...]) in rank order -> [sum B_local, ...].
    `counts` (rows of every rank, known to all ranks -- e.g. from shard_range) makes this ONE collective with
    no host synchronisation; without it the counts are exchanged first (one more collective and a device ->
    host read per call).  Uneven shards are padded to the largest one.  A world of one returns `local` untouched
    unless `force_collective` asks for the collective anyway (a single GPU box exercising the RCCL path)."""
    if world == 1 and not force_collective:
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
