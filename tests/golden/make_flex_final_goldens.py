#!/usr/bin/env python3
"""Freeze the reference's reported energies (main.cpp:339-344) for the flexible-residue case of vina_goldens.npz into
tests/golden/flex_final_goldens.npz: eval_adjusted and eval_intramolecular on the combined model for the case's
conformations.  Run in the build container:  python tests/golden/make_flex_final_goldens.py   (values only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests import ref_cases as RC  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if not ref.available():
        sys.exit("oracle/_ref cannot be built here (needs /root/reference)")
    G = np.load(os.path.join(HERE, "vina_goldens.npz"))
    rigid = open(RC.GSK3B).read()
    lig_text = bytes(G["flex/lig_text"]).decode()
    s = ref.Scene(rigid, lig_text, flex_text=open(RC.FLEX_RES).read())
    s.build_grids(G["flex/center"], G["flex/size"])
    confs = G["flex/confs"]
    fe = [s.final_energies(c) for c in confs]
    y = 100.0 / s.conf_independent(100.0)                       # 1 + w * num_tors / 5 (everything.h:796-814)
    w = 0.1 * ((5 * 0.05846 / 0.1 - 1) + 1)
    out = {"e": np.array([x[0] for x in fe], np.float32), "intra": np.array([x[1] for x in fe], np.float32),
           "num_tors": np.float32((y - 1) * 5 / w)}
    p = os.path.join(HERE, "flex_final_goldens.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, "num_tors", out["num_tors"], "e", out["e"][:4], "intra", out["intra"][:4])


if __name__ == "__main__":
    main()
