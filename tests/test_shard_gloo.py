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


def _worker_bcast(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(5)
    rec_xyz = rng.normal(size=(37, 3)).astype(np.float32)
    rec_smt = rng.randint(0, 28, 37).astype(np.int32)
    got = shard.broadcast_arrays([rec_xyz, rec_smt] if rank == 0 else None, dist)
    ok = np.array_equal(got[0], rec_xyz) and np.array_equal(got[1], rec_smt) and got[1].dtype == np.int32
    # round-robin ligands (config C4): every rank "scores" its ligands, rank order restored by the gather
    n_lig = 11
    mine = shard.round_robin(n_lig, rank, world)
    local = np.stack([mine.astype(np.float32), (mine * mine).astype(np.float32)], 1) if len(mine) else np.zeros((0, 2), np.float32)
    full = shard.gather_round_robin(local, n_lig, dist)
    ok = ok and np.array_equal(full[:, 0], np.arange(n_lig)) and np.array_equal(full[:, 1], np.arange(n_lig) ** 2)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_receptor_broadcast_and_round_robin_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bcast, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r] for r in range(world))


def _worker_gpu(rank, world, port, q):
    """two processes on the visible GPUs over RCCL: the bench's verification pattern"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from gnina_amd import capi, synth
    capi.init(rank)
    m = capi.Model("default2017")
    s = capi.Scorer([m])
    blob = None
    if rank == 0:
        rng = np.random.RandomState(0)
        rx, rs = synth.make_receptor(rng, 1200, synth.mapped_types(m.chan_of_smt(False)))
        lx, ls = synth.make_ligand(rng, 24, synth.mapped_types(m.chan_of_smt(True)))
        blob = [rx, rs, lx, ls]
    rx, rs, lx, ls = shard.broadcast_arrays(blob, dist, dev)
    s.set_receptor(rx, rs)
    poses = synth.make_poses(np.random.RandomState(7), lx, 37)

    def f(p):
        o = s.score_batch(p, ls)
        return np.stack([o["pose"], o["affinity"], o["loss"], o["variance"]], 1)

    got = shard.score_sharded(f, poses, dist, dev)
    if rank == 0:
        q.put(bool(np.array_equal(got, f(poses))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's scaling run exercises bench.py --gpus N)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok
