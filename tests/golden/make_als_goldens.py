#!/usr/bin/env python3
"""Freeze what THE REFERENCE (oracle/_ref) computes with --accurate_line_search (bfgs.h:104-180) into
tests/golden/als_goldens.npz for the GPU test: quasi_newton results after 1 / 3 / all iterations and short
Monte-Carlo chains.  Run in the build container:  python tests/golden/make_als_goldens.py   (values only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests import ref_cases as RC  # noqa: E402
from gnina_amd import capi  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "als_goldens.npz")
V3, HUNT = (1000.0, 1000.0, 1000.0), (10.0, 10.0, 10.0)


def main():
    if not ref.available():
        sys.exit("oracle/_ref cannot be built here (needs /root/reference)")
    rigid = open(RC.GSK3B).read()
    lig_text = RC.cys_adduct_ligand()
    lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
    center, size = RC.box_of(lig["coords0"])
    s = ref.Scene(rigid, lig_text)
    s.set_line_search(True)
    b, e, n = s.build_grids(center, size)
    rx, rs = s.grid_atoms()
    G = {"lig_text": np.frombuffer(lig_text.encode(), dtype=np.uint8), "rec_xyz": rx, "rec_smt": rs, "begin": b, "end": e,
         "n": n, "types": np.array(sorted(set(int(t) for t in lig["smt"] if t > 1)), np.int32)}
    rng = np.random.RandomState(17)
    confs = np.concatenate([RC.random_confs(rng, lig["conf0"], 8, small=True), RC.random_confs(rng, lig["conf0"], 8)])
    G["confs"] = confs
    mi = (25 + s.n_movable) // 3
    G["max_iters"] = np.int32(mi)
    for tag, v in (("v1000", V3), ("v10", HUNT)):
        for iters in (1, 3, mi):
            r = [s.bfgs(c, v, max_iters=iters) for c in confs]
            G[f"bfgs/{tag}/{iters}/e"] = np.array([x[0] for x in r], np.float32)
            G[f"bfgs/{tag}/{iters}/conf"] = np.stack([x[1] for x in r])
    r = [s.bfgs(c, V3, ig=1, max_iters=3) for c in confs]                         # quasi_newton on non_cache
    G["bfgs_noncache/3/e"] = np.array([x[0] for x in r], np.float32)
    G["bfgs_noncache/3/conf"] = np.stack([x[1] for x in r])
    for steps in (1, 3):
        rows = [s.mc(seed_, steps, b, e, max_iters=2, num_saved=20) for seed_ in range(100, 132)]
        G[f"mcshort/{steps}/e0"] = np.array([r[0][0] for r in rows], np.float32)
        G[f"mcshort/{steps}/conf0"] = np.stack([r[1][0] for r in rows])
    # --simple_ascent (minimization_params::Simple: simple_gradient_ascent under the same accurate line search)
    s.set_line_search(False, simple=True)
    for iters in (1, 3, mi):
        r = [s.bfgs(c, HUNT, max_iters=iters) for c in confs]
        G[f"simple/v10/{iters}/e"] = np.array([x[0] for x in r], np.float32)
        G[f"simple/v10/{iters}/conf"] = np.stack([x[1] for x in r])
    rows = [s.mc(seed_, 3, b, e, max_iters=2, num_saved=20) for seed_ in range(100, 132)]
    G["simple/mcshort/3/e0"] = np.array([r[0][0] for r in rows], np.float32)
    G["simple/mcshort/3/conf0"] = np.stack([r[1][0] for r in rows])
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
