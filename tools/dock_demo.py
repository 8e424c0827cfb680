#!/usr/bin/env python3
"""End-to-end docking on the GPU (BASELINE config C3: one synthetic complex, exhaustiveness 64,
Monte-Carlo + BFGS on the Vina cache grids, then refine -> CNN rescore -> exact energies -> rank),
composed only from C-ABI calls -- the sequence gnina's main_procedure / do_search run
(main.cpp:428-510, 210-411).  Prints a JSON line with the wall time of every stage."""
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


def setup_grid_dims(center, size, gran=0.375):   # main.cpp:622-634
    n = np.ceil(np.asarray(size, dtype=np.float32) / np.float32(gran)).astype(np.int32)
    span = np.float32(gran) * n.astype(np.float32)
    begin = np.asarray(center, dtype=np.float32) - span / 2
    return begin, begin + span, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exhaustiveness", type=int, default=64)
    ap.add_argument("--steps", type=int, default=0, help="MC steps per chain (0 = gnina's heuristic)")
    ap.add_argument("--models", default="default2017")
    ap.add_argument("--ligands", type=int, default=1, help="dock this many copies concurrently (screening mode)")
    ap.add_argument("--receptor", default="", help="rigid receptor .pdbqt (default: the synthetic C3 complex)")
    ap.add_argument("--ligand", default="", help="ligand .pdbqt; the search box is its bounding box + 4 A (autobox)")
    ap.add_argument("--out", default="", help="write the ranked poses as a multi-MODEL .pdbqt (needs --ligand)")
    args = ap.parse_args()
    capi.init(0)
    if args.receptor and args.ligand:   # real files through the native PDBQT reader (no OpenBabel)
        rec_xyz, rec_smt = capi.read_pdbqt_receptor(args.receptor)
        lig = capi.read_pdbqt_ligand(args.ligand)
        lo, hi = lig["coords0"].min(0) - 4.0, lig["coords0"].max(0) + 4.0   # --autobox_ligand, autobox_add 4 (main.cpp:1470-1490)
        sc = {"rec_xyz": rec_xyz, "rec_smt": rec_smt, "lig": lig, "center": (lo + hi) / 2, "size": hi - lo}
    else:
        sc = vina_scene.build(0)
        lig = sc["lig"]
    T = lig["n_tors"]
    begin, end, n = setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    t = {}
    t0 = time.perf_counter()
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    vina.build_cache(begin, end, n, types, 1e3)
    vina.set_ligand(lig)
    t["tables_cache_setup_s"] = time.perf_counter() - t0
    n_mov = len(lig["smt"])
    heuristic = n_mov + 10 * (6 + T)                      # main.cpp:441-443
    steps = args.steps or int(70 * 3 * (50 + heuristic) / 2)
    iters = (25 + n_mov) // 3
    P = capi.McParams.default(steps, iters, 50)
    seeds = np.arange(1, args.exhaustiveness * args.ligands + 1, dtype=np.uint64) * np.uint64(7919)
    t0 = time.perf_counter()
    cnt, e, cf, xyz, ev = vina.mc_batch(seeds, begin, end, P)
    t["mc_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    me, mcf, mxyz = capi.merge_mc_outputs(cnt[:args.exhaustiveness], e[:args.exhaustiveness], cf[:args.exhaustiveness],
                                          xyz[:args.exhaustiveness], 2.0, 50)
    er, rcf, tries = vina.refine_batch(mcf)
    t["merge_refine_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    scorer = capi.Scorer(args.models.split(","))
    scorer.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    t["cnn_setup_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, co = vina.eval_batch(rcf, want_coords=True)
    out = scorer.score_batch(co, lig["smt"])
    ef, intra = vina.final_energies(rcf, float(T))
    heavy = np.nonzero(lig["smt"] > 1)[0]
    keep = capi.rank_poses(out["pose"], out["affinity"], ef, co[:, heavy], 0, 1.0)
    t["cnn_rescore_rank_s"] = time.perf_counter() - t0
    res = {"config": f"C3: exhaustiveness {args.exhaustiveness} x {args.ligands} ligand(s), {steps} MC steps/chain, "
                     f"{n_mov}-atom ligand with {T} torsions, receptor {len(sc['rec_smt'])} atoms, models {args.models}",
           "stages_s": {k: round(v, 4) for k, v in t.items()},
           "total_s": round(sum(t.values()), 3),
           "mc_evals": int(ev.sum()), "mc_evals_per_s": round(float(ev.sum()) / t["mc_s"]),
           "poses_merged": int(len(me)), "poses_reported": int(len(keep)),
           "best": {"cnnscore": float(out["pose"][keep[0]]), "cnnaffinity": float(out["affinity"][keep[0]]),
                    "vina_affinity": float(ef[keep[0]]), "intramol": float(intra[keep[0]])}}
    if args.out and args.ligand:
        order = list(keep)
        text = capi.pdbqt_poses_text(args.ligand, co[order], ef[order], out["pose"][order], out["affinity"][order])
        open(args.out, "w").write(text)
        res["written"] = {"file": args.out, "poses": len(order)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
