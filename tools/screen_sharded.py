#!/usr/bin/env python3
"""BASELINE config C4 as a job: 1 receptor x N ligands through the crossdock_default2018 ensemble, ligands dealt
round-robin to one process per GPU, no data-path collective -- one receptor broadcast at the start, one
all_gather of the scores at the end (RCCL over xGMI; 16 B per pose).

    python tools/screen_sharded.py --ligands 4096                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/screen_sharded.py --ligands 100000                        # the 8-GPU job

Rank 0 prints one JSON line: ligands/s over the whole job and a checksum of the gathered scores that does not depend
on the number of ranks (every ligand's scores are the same bits wherever it was scored)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ligand(seed, lig_types, synth, poses=9, lmax=48):
    """ligand `seed` of the screen: L ~ U{16..48} atoms, 9 poses; a function of the seed only"""
    rng = np.random.RandomState(1000003 + seed)
    L = rng.randint(16, lmax + 1)
    lx, ls = synth.make_ligand(rng, L, lig_types)
    xyz = np.zeros((poses, lmax, 3), dtype=np.float32)
    smt = np.full((poses, lmax), -1, dtype=np.int32)
    xyz[:, :L] = synth.make_poses(rng, lx, poses)
    smt[:, :L] = ls
    return xyz, smt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ligands", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=1024, help="ligands per ragged batch")
    ap.add_argument("--models", default="crossdock_default2018_ensemble")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from gnina_amd import capi, shard, synth
    capi.init(local_rank)
    wdir = os.path.join(ROOT, "gnina_amd", "weights")
    if args.models.endswith("_ensemble"):   # prefix expansion, cnn_torch_scorer.cpp:52-64
        prefix = args.models[:-len("_ensemble")]
        names = sorted(f[:-4] for f in os.listdir(wdir) if f.endswith(".mgw") and f.startswith(prefix))
    else:
        names = args.models.split(",")
    models = [capi.Model(n) for n in names]
    scorer = capi.Scorer(models)
    rec_types = synth.mapped_types(models[0].chan_of_smt(False))
    lig_types = synth.mapped_types(models[0].chan_of_smt(True))
    blob = list(synth.make_receptor(np.random.RandomState(0), 2500, rec_types)) if rank == 0 else None
    rec_xyz, rec_smt = shard.broadcast_arrays(blob, dist, dev)
    scorer.set_receptor(rec_xyz, rec_smt)
    mine = shard.round_robin(args.ligands, rank, world)
    P = 9
    # the screen's input: generated before the clock starts (a real job reads prepared ligand files; making synthetic
    # molecules is not part of it)
    batches = []
    for b0 in range(0, len(mine), args.batch):
        ids = mine[b0:b0 + args.batch]
        lig = [ligand(int(i), lig_types, synth, P) for i in ids]
        batches.append((b0, len(ids), np.concatenate([l[0] for l in lig]), np.concatenate([l[1] for l in lig])))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local = np.zeros((len(mine), 3 * P), dtype=np.float32)          # per ligand: 9 x (pose, affinity, variance)
    for b0, nb, xyz, smt in batches:
        ids = mine[b0:b0 + nb]
        o = scorer.score_ragged(xyz, smt)
        local[b0:b0 + len(ids)] = np.stack([o["pose"], o["affinity"], o["variance"]], 1).reshape(len(ids), P, 3).reshape(len(ids), -1)
    full = shard.gather_round_robin(local, args.ligands, dist, dev)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        best = full.reshape(args.ligands, P, 3)[:, :, 0].max(1)
        print(json.dumps({"config": f"C4 job: {args.ligands} ligands x {P} poses, {len(names)} models ({args.models}), "
                                    f"round-robin over {world} rank(s)",
                          "n_gpus": world, "seconds": round(dt, 3), "ligands_per_s": round(args.ligands / dt, 1),
                          "poses_per_s": round(args.ligands * P / dt, 1),
                          "note": "ligands prepared before the timed region; host pointers, PCIe and per-call set-up inside it",
                          "checksum": float(np.float64(full.astype(np.float64).sum())),
                          "best_cnnscore_mean": float(best.mean())}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
