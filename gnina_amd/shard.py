"""Pose / ligand sharding across one-process-per-GPU ranks.

The path shards naturally: poses are independent, nothing is reduced across them.  Receptor, type
tables and weights are replicated per rank; the batch is split contiguously; each rank scores its
shard with no data-path collective; the only exchange is one gather of B x {pose, affinity, loss,
variance} floats to rank 0 at the end (16 B per pose: latency-bound on xGMI, so a single
all_gather on equal-size padded shards -- never a ring all-reduce)."""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous [begin, end) of rank's shard; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def score_sharded(score_fn, lig_xyz, dist=None, device=None):
    """Score poses [B, L, 3] across all ranks.

    score_fn(poses_shard) -> float32 array [n_shard, K] (K outputs per pose) on this rank.
    Returns the full [B, K] array on every rank (all_gather over RCCL on GPUs, gloo on CPU).
    With dist=None (single process) it is just score_fn(lig_xyz)."""
    import torch
    B = len(lig_xyz)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(score_fn(lig_xyz), dtype=np.float32)
    rank, world = dist.get_rank(), dist.get_world_size()
    b0, b1 = shard_range(B, rank, world)
    local = np.asarray(score_fn(lig_xyz[b0:b1]), dtype=np.float32) if b1 > b0 else np.zeros((0, 0), np.float32)
    if b1 > b0:
        local = local.reshape(b1 - b0, -1)
    K = local.shape[1] if b1 > b0 else None
    k_t = torch.tensor([K if K is not None else 0], dtype=torch.int64, device=device)
    dist.all_reduce(k_t, op=dist.ReduceOp.MAX)
    K = int(k_t.item())
    pad = max(shard_sizes(B, world))
    buf = torch.zeros(pad, K, dtype=torch.float32, device=device)
    if b1 > b0:
        buf[: b1 - b0] = torch.from_numpy(local.reshape(b1 - b0, K)).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    sizes = shard_sizes(B, world)
    return np.concatenate([out[r][: sizes[r]].cpu().numpy() for r in range(world)], axis=0)
