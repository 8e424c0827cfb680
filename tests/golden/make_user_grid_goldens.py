#!/usr/bin/env python3
"""Freeze what THE REFERENCE (oracle/_ref) computes with a --user_grid into tests/golden/user_grid_goldens.npz for the
GPU test (tests/test_gpu_vina_ref.py::test_user_grid_*): the cache lattice with the grid baked in (cache.cpp:177-179),
non_cache::eval_deriv with the per-atom term (non_cache.cpp:168-173), model::eval with its own sum over the ligand's
atoms (model.cu:125-134) on both igrids, and the final energies.  Run in the build container:
    python tests/golden/make_user_grid_goldens.py        (values only -- no reference source is copied)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests import ref_cases as RC  # noqa: E402
from gnina_amd import capi  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "user_grid_goldens.npz")
V3 = (1000.0, 1000.0, 1000.0)


def main():
    if not ref.available():
        sys.exit("oracle/_ref cannot be built here (needs /root/reference)")
    rigid = open(RC.GSK3B).read()
    lig_text = RC.cys_adduct_ligand()
    lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
    center, size = RC.box_of(lig["coords0"])
    text, value_lines = RC.user_grid_text(center, (18, 20, 16), 0.75)
    ub, ue, un, vals = capi.user_grid_parse(text)
    scale = np.float32(0.75)
    s = ref.Scene(rigid, lig_text)
    s.set_user_grid(ub, ue, un, value_lines, scale)
    b, e, n = s.build_grids(center, size)
    rx, rs = s.grid_atoms()
    G = {"lig_text": np.frombuffer(lig_text.encode(), dtype=np.uint8),
         "user_grid_text": np.frombuffer(text.encode(), dtype=np.uint8), "scale": scale,
         "ub": ub, "ue": ue, "un": un, "rec_xyz": rx, "rec_smt": rs, "center": np.asarray(center, np.float32),
         "size": np.asarray(size, np.float32), "begin": b, "end": e, "n": n}
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    G["types"] = np.array(types, np.int32)
    rng = np.random.RandomState(11)
    idx = rng.randint(0, [n[0] + 1, n[1] + 1, n[2] + 1], size=(800, 3)).astype(np.int32)
    pts = np.stack([b[i] + (e[i] - b[i]) * idx[:, i].astype(np.float32) / np.float32(n[i]) for i in range(3)], 1)
    G["grid_idx"] = idx
    G["grid_val"] = np.stack([s.cache_probe(t, pts.astype(np.float32), v=3.4e38) for t in types])
    confs = np.concatenate([RC.random_confs(rng, lig["conf0"], 6, small=True), RC.random_confs(rng, lig["conf0"], 4)])
    G["confs"] = confs
    r = [s.eval_deriv(c, V3, ig=1) for c in confs]
    G["noncache/e"] = np.array([x[0] for x in r], np.float32)
    G["noncache/change"] = np.stack([x[1] for x in r])
    G["noncache/eval"] = np.array([s.eval(c, V3, ig=1) for c in confs], np.float32)
    r = [s.eval_deriv(c, V3) for c in confs]
    G["cache/e"] = np.array([x[0] for x in r], np.float32)
    G["cache/change"] = np.stack([x[1] for x in r])
    G["cache/eval"] = np.array([s.eval(c, V3) for c in confs], np.float32)
    fe = [s.final_energies(c) for c in confs]
    G["final/e"] = np.array([x[0] for x in fe], np.float32)
    G["final/intra"] = np.array([x[1] for x in fe], np.float32)
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
