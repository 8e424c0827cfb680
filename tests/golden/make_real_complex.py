"""Freeze the reference's own real-complex inputs as a fixture for the GPU box (which has no /root/reference):
the GSK3B receptor (3,460 atoms with polar hydrogens) and the flexible-residue file of test/gnina/data, as the PDBQT
texts both stacks parse, plus the two ligands tests/ref_cases.py derives from them.  bench.py's `also.real_complex` and
`also.c3_real` entries and the flexible-residue CNN tests read it.
    python tests/golden/make_real_complex.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import ref_cases  # noqa: E402


def text(s):
    return np.frombuffer(s.encode(), dtype=np.uint8)


np.savez_compressed(os.path.join(ROOT, "tests", "golden", "real_complex.npz"),
                    rec_pdbqt=text(open(ref_cases.GSK3B).read()),
                    flex_pdbqt=text(open(ref_cases.FLEX_RES).read()),
                    lig_adduct_pdbqt=text(ref_cases.cys_adduct_ligand()),
                    lig_chain_pdbqt=text(ref_cases.long_chain_ligand()),
                    source=text("test/gnina/data/GSK3B_DFG_out_35-388-processed_rigid.pdbqt, flex_res_side_chain.pdbqt"))
