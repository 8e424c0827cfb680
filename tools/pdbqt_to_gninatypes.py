#!/usr/bin/env python3
"""gninatyper for PDBQT inputs (gninatyper.cpp:30-36,65-75 writes (x, y, z, smina type) records): type a rigid receptor,
a receptor with flexible residues, or a ligand with the native PDBQT reader and write a .gninatypes file.

    python tools/pdbqt_to_gninatypes.py rec.pdbqt rec.gninatypes
    python tools/pdbqt_to_gninatypes.py --ligand lig.pdbqt lig.gninatypes
    python tools/pdbqt_to_gninatypes.py --flex flex.pdbqt rigid.pdbqt rec.gninatypes   # movable | inflex | rigid rows
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnina_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--ligand", action="store_true", help="the input is a ligand (ROOT / BRANCH / TORSDOF)")
    ap.add_argument("--flex", default="", help="flexible residues (.pdbqt with BEGIN_RES blocks) of the receptor")
    a = ap.parse_args()
    if a.ligand:
        lig = capi.read_pdbqt_ligand(a.input)
        xyz, smt = lig["coords0"], lig["smt"]
    elif a.flex:
        xyz, smt, n_mov, n_inflex = capi.read_pdbqt_receptor_flex(a.input, a.flex)
        print(f"{n_mov} movable + {n_inflex} inflex + {len(smt) - n_mov - n_inflex} rigid atoms")
    else:
        xyz, smt = capi.read_pdbqt_receptor(a.input)
    capi.write_gninatypes(a.output, xyz, smt)
    print(f"{len(smt)} atoms -> {a.output}")


if __name__ == "__main__":
    main()
