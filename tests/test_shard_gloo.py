"""The N>1 path on CPU: world_size-2 (and 3) gloo processes run gnina_amd.shard.score_sharded with a
deterministic stand-in scoring function; the gathered result must equal the single-process result
in pose order, including ragged shards and an empty shard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnina_amd import shard  # noqa: E402


def fake_score(poses):
    """deterministic per-pose function: 4 outputs like {pose, affinity, loss, variance}"""
    p = np.asarray(poses, dtype=np.float64)
    s = p.reshape(len(p), -1)
    return np.stack([s.sum(1), (s ** 2).sum(1), s.min(1, initial=0.0), s.max(1, initial=0.0)], axis=1).astype(np.float32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)
    poses = rng.normal(size=(B, 5, 3)).astype(np.float32)
    out = shard.score_sharded(fake_score, poses, dist)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 8, 1024, 100003):
        for w in (1, 2, 3, 8):
            rs = [shard.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,B", [(2, 11), (3, 2)])
def test_score_sharded_gloo_matches_single_process(world, B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.RandomState(0)
    poses = rng.normal(size=(B, 5, 3)).astype(np.float32)
    assert np.array_equal(out, fake_score(poses))
