#!/usr/bin/env python3
"""Freeze what THE REFERENCE (oracle/_ref) computes under --approximation spline (precalculate_splines(sf, 10): gnina's
default for --minimize) into tests/golden/spline_goldens.npz for the GPU test: spline samples, cache lattice samples,
model::eval_deriv / eval on cache and non_cache, quasi_newton results.  Run in the build container:
    python tests/golden/make_spline_goldens.py        (values only -- no reference source is copied)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests import ref_cases as RC  # noqa: E402
from gnina_amd import capi  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spline_goldens.npz")
V3, HUNT = (1000.0, 1000.0, 1000.0), (10.0, 10.0, 10.0)


def main():
    if not ref.available():
        sys.exit("oracle/_ref cannot be built here (needs /root/reference)")
    rigid = open(RC.GSK3B).read()
    lig_text = RC.cys_adduct_ligand()
    lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
    center, size = RC.box_of(lig["coords0"])
    s = ref.Scene(rigid, lig_text)
    s.set_approximation(1, 10.0)
    b, e, n = s.build_grids(center, size)
    rx, rs = s.grid_atoms()
    G = {"lig_text": np.frombuffer(lig_text.encode(), dtype=np.uint8), "factor": np.float32(10.0), "rec_xyz": rx,
         "rec_smt": rs, "begin": b, "end": e, "n": n}
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    G["types"] = np.array(types, np.int32)
    rng = np.random.RandomState(13)
    r2 = np.concatenate([np.linspace(0.05, 63.9, 300), rng.uniform(0.3, 64, 200), [63.99, 64.0, 70.0]]).astype(np.float32)
    pairs = [(2, 2), (2, 13), (7, 13), (10, 4), (3, 8), (23, 12)]
    G["sp/r2"], G["sp/pairs"] = r2, np.array(pairs, np.int32)
    t = [s.prec_eval(a, c, r2) for a, c in pairs]
    G["sp/e"], G["sp/dor"] = np.stack([x[0] for x in t]), np.stack([x[1] for x in t])
    idx = rng.randint(0, [n[0] + 1, n[1] + 1, n[2] + 1], size=(600, 3)).astype(np.int32)
    pts = np.stack([b[i] + (e[i] - b[i]) * idx[:, i].astype(np.float32) / np.float32(n[i]) for i in range(3)], 1)
    G["grid_idx"] = idx
    G["grid_val"] = np.stack([s.cache_probe(tt, pts.astype(np.float32), v=3.4e38) for tt in types])
    confs = np.concatenate([RC.random_confs(rng, lig["conf0"], 6, small=True), RC.random_confs(rng, lig["conf0"], 6)])
    G["confs"] = confs
    for tag, v in (("v1000", V3), ("v10", HUNT)):
        r = [s.eval_deriv(c, v) for c in confs]
        G[tag + "/e"] = np.array([x[0] for x in r], np.float32)
        G[tag + "/change"] = np.stack([x[1] for x in r])
        G[tag + "/eval"] = np.array([s.eval(c, v) for c in confs], np.float32)
    r = [s.eval_deriv(c, V3, ig=1) for c in confs]
    G["noncache/e"] = np.array([x[0] for x in r], np.float32)
    G["noncache/change"] = np.stack([x[1] for x in r])
    G["noncache/eval"] = np.array([s.eval(c, V3, ig=1) for c in confs], np.float32)
    mi = (25 + s.n_movable) // 3
    G["max_iters"] = np.int32(mi)
    for iters in (1, 3, mi):
        r = [s.bfgs(c, V3, max_iters=iters) for c in confs]
        G[f"bfgs/{iters}/e"] = np.array([x[0] for x in r], np.float32)
        G[f"bfgs/{iters}/conf"] = np.stack([x[1] for x in r])
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
