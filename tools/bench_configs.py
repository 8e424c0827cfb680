#!/usr/bin/env python3
"""Measurements of the BASELINE configurations that are not bench.py's headline line:
  C4  virtual screen: 15-model crossdock_default2018 ensemble, 9 poses per ligand, L ~ U{16..48}, ragged batches
  C5  dense_1_3 at 0.25 A / 96^3: forward, forward+backward, and CNN refinement evaluations
  plus gnina's default 3-model ensemble at 48^3 (forward and CNN refinement).
One JSON object per line.  Everything goes through the C ABI with host pointers (PCIe and host set-up included)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    capi.init(0)
    rng = np.random.RandomState(0)
    m0 = capi.Model("crossdock_default2018")
    rec_types, lig_types = synth.mapped_types(m0.chan_of_smt(False)), synth.mapped_types(m0.chan_of_smt(True))
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, rec_types)

    def want(k):
        return not args.only or k in args.only.split(",")

    if want("c4"):
        # weights: 15 handles of the shipped crossdock_default2018 blob (the ensemble's members share the
        # architecture; only 3 of the 15 blobs are committed, the timing does not depend on the values)
        s = capi.Scorer([capi.Model("crossdock_default2018") for _ in range(15)])
        s.set_receptor(rec_xyz, rec_smt)
        n_lig, P, Lmax = 1024, 9, 48
        xyz = np.zeros((n_lig * P, Lmax, 3), dtype=np.float32)
        smt = np.full((n_lig * P, Lmax), -1, dtype=np.int32)
        for i in range(n_lig):
            L = rng.randint(16, 49)
            lx, ls = synth.make_ligand(rng, L, lig_types)
            xyz[i * P:(i + 1) * P, :L] = synth.make_poses(rng, lx, P)
            smt[i * P:(i + 1) * P, :L] = ls
        dt = timed(lambda: s.score_ragged(xyz, smt), args.reps)
        print(json.dumps({"config": "C4 virtual screen, 15 x Default2018, 9 poses/ligand, ragged batch of 1024 ligands",
                          "ligands_per_s": n_lig / dt, "poses_per_s": n_lig * P / dt,
                          "model_forwards_per_s": 15 * n_lig * P / dt, "s_per_100k_ligands_1gpu": 1e5 / (n_lig / dt)}),
              flush=True)
        del s
    lx, ls = synth.make_ligand(rng, 32, lig_types)
    if want("ens"):
        s = capi.Scorer(["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"])
        s.set_receptor(rec_xyz, rec_smt)
        poses = synth.make_poses(rng, lx, 1024)
        dt = timed(lambda: s.score_batch(poses, ls), args.reps)
        dg = timed(lambda: s.score_grad(poses, ls), args.reps)
        print(json.dumps({"config": "default ensemble (dense_1_3, dense_1_3_PT_KD_3, crossdock_default2018_KD_4), 48^3, B=1024",
                          "poses_per_s_forward": 1024 / dt, "poses_per_s_forward_backward": 1024 / dg}), flush=True)
        del s
    if want("c5"):
        m = capi.Model("dense_1_3", resolution=0.25, dimension=23.75)
        s = capi.Scorer([m])
        s.set_receptor(rec_xyz, rec_smt)
        B = 256
        poses = synth.make_poses(rng, lx, B)
        dt = timed(lambda: s.score_batch(poses, ls), args.reps)
        dg = timed(lambda: s.score_grad(poses, ls), args.reps)
        gf = 36.33e9
        print(json.dumps({"config": "C5 dense_1_3 @ 0.25 A (96^3), fp32, B=256", "poses_per_s_forward": B / dt,
                          "tflops_forward": B / dt * gf / 1e12, "poses_per_s_forward_backward": B / dg}), flush=True)
        s.set_precision(True)
        dt = timed(lambda: s.score_batch(poses, ls), args.reps)
        dg = timed(lambda: s.score_grad(poses, ls), args.reps)
        print(json.dumps({"config": "C5 dense_1_3 @ 0.25 A (96^3), bf16 MFMA, B=256", "poses_per_s_forward": B / dt,
                          "tflops_forward": B / dt * gf / 1e12, "poses_per_s_forward_backward": B / dg}), flush=True)
        del s
    if want("refine"):
        from gnina_amd import vina_scene
        sc = vina_scene.build(seed=3)
        lig = sc["lig"]
        v = capi.Vina()
        v.set_ligand(lig)
        lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
        for label, models, B, bf in (("default ensemble 48^3", ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"], 64, False),
                                     ("default2017 48^3", ["default2017"], 64, False),
                                     ("dense_1_3 96^3", [capi.Model("dense_1_3", resolution=0.25, dimension=23.75)], 64, False),
                                     ("dense_1_3 96^3 bf16", [capi.Model("dense_1_3", resolution=0.25, dimension=23.75)], 64, True)):
            s = capi.Scorer(models)
            s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
            s.set_precision(bf)
            dim = 23.75 if "96" in label else 23.5
            box = capi.CnnBox.make(dim, lo, hi)
            r2 = np.random.RandomState(1)
            confs = np.stack([lig["conf0"]] * B).astype(np.float32)
            confs[:, :3] += r2.uniform(-0.5, 0.5, (B, 3)).astype(np.float32)
            confs[:, 7:] += r2.uniform(-0.4, 0.4, confs[:, 7:].shape).astype(np.float32)
            v.cnn_refine_batch(s, confs[:4], box, max_iters=1)   # warm-up / allocation
            t0 = time.perf_counter()
            e, out, tries, evals = v.cnn_refine_batch(s, confs, box)
            dt = time.perf_counter() - t0
            start, _ = v.cnn_eval_batch(s, confs, box, None, deriv=False)
            print(json.dumps({"config": f"CNN refinement (refine_structure on non_cache_cnn), {label}, {B} poses",
                              "seconds": dt, "cnn_evals": int(evals.sum()), "cnn_evals_per_s": float(evals.sum() / dt),
                              "mean_loss_start": float(start.mean()), "mean_loss_end": float(e.mean()),
                              "s_per_pose": dt / B}), flush=True)
            del s


if __name__ == "__main__":
    main()
