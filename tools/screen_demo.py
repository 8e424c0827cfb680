#!/usr/bin/env python3
"""Docking screen with DIFFERENT ligands in flight: N synthetic ligands (20-40 atoms, 3-9 torsions) against one
receptor, each docked like gnina's default run (exhaustiveness 8, main.cpp:441-463 step heuristic): the chains of
ALL ligands run in one launch (mi_vina_mc_screen; streams of separate handles only overlap as far as the runtime
has hardware queues -- measured: about four kernels at a time), then per ligand merge -> refine -> final
energies, and the kept poses of all ligands are CNN-rescored in one ragged batch.

    python tools/screen_demo.py --ligands 256

Prints one JSON line: ligands/s docked, evaluations/s, rescoring time."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402
from gnina_amd import vina_scene  # noqa: E402
from tools.dock_demo import setup_grid_dims  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ligands", type=int, default=96)
    ap.add_argument("--exhaustiveness", type=int, default=8)
    ap.add_argument("--steps", type=int, default=0, help="MC steps per chain (0 = gnina's heuristic)")
    ap.add_argument("--models", default="default2017")
    ap.add_argument("--poses", type=int, default=9, help="poses kept per ligand (num_modes)")
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # streams -> hardware queues (mi_gnina_init does the same)
    capi.init(0)
    sc = vina_scene.build(0)                      # receptor + pocket of the C3 complex
    rng = np.random.RandomState(7)
    ligs = []
    for i in range(args.ligands):
        lig = synth.make_ligand_tree(rng, int(rng.randint(20, 41)), int(rng.randint(3, 10)))
        shift = -lig["coords0"].mean(0)
        lig["coords0"] = (lig["coords0"] + shift).astype(np.float32)
        lig["conf0"][:3] += shift
        ligs.append(lig)
    size = np.full(3, 22.0, np.float32)            # one search box for the whole screen (the binding site)
    begin, end, n = setup_grid_dims(np.zeros(3, np.float32), size)
    types = sorted({int(t) for lig in ligs for t in lig["smt"] if t > 1})
    t0 = time.perf_counter()
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    vina.build_cache(begin, end, n, types, 1e3)
    vina.set_screen(ligs)
    params, chain_lig, seeds = [], [], []
    for i, lig in enumerate(ligs):
        T, n_mov = lig["n_tors"], len(lig["smt"])
        steps = args.steps or int(70 * 3 * (50 + n_mov + 10 * (6 + T)) / 2)      # main.cpp:441-443
        params.append(capi.McParams.default(steps, (25 + n_mov) // 3, 50))
        chain_lig += [i] * args.exhaustiveness
        seeds += [(1000 * i + k + 1) * 7919 for k in range(args.exhaustiveness)]
    t_setup = time.perf_counter() - t0
    # every chain of every ligand in ONE launch (mi_vina_mc_screen)
    t0 = time.perf_counter()
    cnt, e, cf, xyz, ev = vina.mc_screen(np.array(chain_lig, np.int32), np.array(seeds, np.uint64), begin, end, params)
    t_mc = time.perf_counter() - t0
    # merge each ligand's chain containers (host), then refine / coordinates / final energies of the kept poses of
    # ALL ligands in one launch each (mi_vina_*_screen)
    t0 = time.perf_counter()
    E = args.exhaustiveness
    item, rows = [], []
    for i, lig in enumerate(ligs):
        nc, nh = 7 + lig["n_tors"], int((lig["smt"] > 1).sum())
        sl = slice(i * E, (i + 1) * E)
        me, mcf, mxyz = capi.merge_mc_outputs(cnt[sl], e[sl], np.ascontiguousarray(cf[sl][:, :, :nc]),
                                              np.ascontiguousarray(xyz[sl][:, :, :3 * nh]).reshape(E, -1, nh, 3), 2.0, 50)
        for c in mcf[:args.poses]:
            r = np.zeros(vina.screen_conf, np.float32)
            r[:nc] = c
            rows.append(r)
            item.append(i)
    item, rows = np.array(item, np.int32), np.stack(rows)
    er, rcf, tries = vina.refine_screen(item, rows, [(25 + len(l["smt"])) // 3 for l in ligs])
    _, _, co_all = vina.eval_screen(item, rcf, deriv=False, want_coords=True)
    ef_all, intra = vina.final_energies_screen(item, rcf, [float(l["n_tors"]) for l in ligs])
    results = []
    for i, lig in enumerate(ligs):
        idx = np.nonzero(item == i)[0]
        results.append((co_all[idx][:, :len(lig["smt"])], ef_all[idx]))
    t_post = time.perf_counter() - t0
    t_dock = t_setup + t_mc + t_post
    evals = [int(ev.sum())]
    # CNN rescoring of every kept pose of every ligand in one ragged batch (mi_scorer_score_ragged)
    t0 = time.perf_counter()
    scorer = capi.Scorer(args.models.split(","))
    scorer.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    lmax = max(len(l["smt"]) for l in ligs)
    xyz, smt, owner = [], [], []
    for i, (co, ef) in enumerate(results):
        for p in range(len(co)):
            x = np.zeros((lmax, 3), np.float32)
            s = np.full(lmax, -1, np.int32)
            x[:co.shape[1]] = co[p]
            s[:co.shape[1]] = ligs[i]["smt"]
            xyz.append(x), smt.append(s), owner.append(i)
    out = scorer.score_ragged(np.stack(xyz), np.stack(smt))
    t_cnn = time.perf_counter() - t0
    best = {}
    for k, i in enumerate(owner):
        if i not in best or out["pose"][k] > out["pose"][best[i]]:
            best[i] = k
    print(json.dumps({
        "config": f"{len(ligs)} different ligands (20-40 atoms, 3-9 torsions), exhaustiveness {args.exhaustiveness} = "
                  f"{len(chain_lig)} chains in one launch (mi_vina_mc_screen), receptor {len(sc['rec_smt'])} atoms",
        "waves_per_chain": os.environ.get("MI_VINA_MC_WAVES", f"auto ({1 if len(chain_lig) > 512 else 2 if len(chain_lig) > 256 else 4})"),
        "setup_s": round(t_setup, 3), "mc_s": round(t_mc, 2), "merge_refine_energies_s": round(t_post, 2),
        "dock_s": round(t_dock, 2), "ligands_per_s": round(len(ligs) / t_dock, 2),
        "mc_evals": int(sum(evals)), "mc_evals_per_s": round(sum(evals) / t_mc),
        "poses_rescored": len(owner), "cnn_rescore_s": round(t_cnn, 3),
        "mean_best_cnnscore": round(float(np.mean([out["pose"][k] for k in best.values()])), 4)}))


if __name__ == "__main__":
    main()
