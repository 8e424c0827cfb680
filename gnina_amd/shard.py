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


def broadcast_arrays(arrays, dist=None, device=None, src=0):
    """One broadcast of a few numpy arrays from rank `src` (the receptor / weight blob hand-out of SURVEY 8e:
    ncclBroadcast at start).  `arrays` is a list on `src` and may be None elsewhere; returns the list on every rank.
    Shapes / dtypes travel first in a small int64 header, the payload as one uint8 tensor: two collectives in all."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [np.asarray(a) for a in arrays]
    rank = dist.get_rank()
    kinds = {"float32": 0, "int32": 1, "int64": 2, "uint8": 3, "float64": 4}
    names = {v: k for k, v in kinds.items()}
    MAXA, MAXD = 8, 4
    head = torch.zeros(1 + MAXA * (2 + MAXD), dtype=torch.int64, device=device)
    if rank == src:
        arrays = [np.ascontiguousarray(a) for a in arrays]
        assert len(arrays) <= MAXA and all(a.ndim <= MAXD for a in arrays)
        h = [len(arrays)]
        for a in arrays:
            h += [kinds[str(a.dtype)], a.ndim] + list(a.shape) + [0] * (MAXD - a.ndim)
        h += [0] * (len(head) - len(h))
        head.copy_(torch.tensor(h, dtype=torch.int64))
    dist.broadcast(head, src=src)
    h = head.cpu().tolist()
    metas = []
    for i in range(h[0]):
        o = 1 + i * (2 + MAXD)
        metas.append((names[h[o]], tuple(h[o + 2:o + 2 + h[o + 1]])))
    nbytes = [int(np.prod(sh, dtype=np.int64)) * np.dtype(dt).itemsize for dt, sh in metas]
    payload = torch.empty(sum(nbytes), dtype=torch.uint8, device=device)
    if rank == src:
        payload.copy_(torch.from_numpy(np.concatenate([a.view(np.uint8).reshape(-1) for a in arrays])))
    dist.broadcast(payload, src=src)
    raw = payload.cpu().numpy()
    out, o = [], 0
    for (dt, sh), nb in zip(metas, nbytes):
        out.append(raw[o:o + nb].view(dt).reshape(sh).copy())
        o += nb
    return out


def round_robin(n_items, rank, world):
    """Ligand l goes to rank l % world: indices of this rank's items."""
    return np.arange(rank, n_items, world)


def gather_round_robin(local, n_items, dist=None, device=None):
    """local [n_local, K] = results of this rank's round-robin items -> [n_items, K] in item order on every rank
    (one all_gather on equal, padded shards)."""
    import torch
    local = np.asarray(local, dtype=np.float32)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.reshape(n_items, -1)
    rank, world = dist.get_rank(), dist.get_world_size()
    K = local.shape[1] if local.ndim == 2 and len(local) else 0
    k_t = torch.tensor([K], dtype=torch.int64, device=device)
    dist.all_reduce(k_t, op=dist.ReduceOp.MAX)
    K = int(k_t.item())
    pad = (n_items + world - 1) // world
    buf = torch.zeros(pad, K, dtype=torch.float32, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(local.reshape(len(local), K)).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    full = np.zeros((n_items, K), dtype=np.float32)
    for r in range(world):
        idx = round_robin(n_items, r, world)
        full[idx] = out[r][:len(idx)].cpu().numpy()
    return full
