"""Writes a synthetic rigid receptor and a small ligand as .pdbqt files (tools/experiments/_gen/, git-ignored) so that
tools/dock_demo.py --receptor/--ligand/--out can be exercised on a GPU box without any reference data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import synth  # noqa: E402
from tests.test_pdbqt_cpu import atom_line, chain_pdbqt  # noqa: E402

AD = {2: "C", 4: "A", 6: "N", 9: "NA", 10: "O", 13: "OA", 14: "S"}
out = os.path.join(ROOT, "tools", "scratch", "_gen")
os.makedirs(out, exist_ok=True)
rng = np.random.RandomState(0)
xyz, smt = synth.make_receptor(rng, 2500, np.array(sorted(AD), dtype=np.int32))
with open(os.path.join(out, "demo_rec.pdbqt"), "w") as f:
    for i in range(len(smt)):
        f.write(atom_line(i + 1, "X", *xyz[i], AD[int(smt[i])]) + "\n")
    f.write("TER\n")
open(os.path.join(out, "demo_lig.pdbqt"), "w").write(chain_pdbqt())
print("wrote", out)
