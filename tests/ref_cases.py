"""Inputs shared by the tests that pin our Vina path to the reference itself (oracle/_ref) and by the script that
freezes reference outputs into tests/golden/vina_goldens.npz.  PDBQT texts only: both stacks parse the same bytes."""
import os

import numpy as np

REF_DATA = "/root/reference/test/gnina/data"
GSK3B = os.path.join(REF_DATA, "GSK3B_DFG_out_35-388-processed_rigid.pdbqt")
FLEX_RES = os.path.join(REF_DATA, "flex_res_side_chain.pdbqt")


def atom_line(serial, name, x, y, z, adtype, q=0.0, het=False, res="LIG", resnum=1):
    rec = "HETATM" if het else "ATOM  "
    return f"{rec}{serial:5d} {name:<4s} {res} A{resnum:4d}    {x:8.3f}{y:8.3f}{z:8.3f}{1.0:6.2f}{0.0:6.2f}    {q:6.3f} {adtype:<2s}"


def cys_adduct_ligand():
    """The reference's flexible-residue fixture (a 32-atom, 10-torsion covalent adduct) re-labelled as a ligand."""
    lines = [l for l in open(FLEX_RES).read().splitlines() if not l.startswith(("BEGIN_RES", "END_RES"))]
    return "\n".join(lines + ["TORSDOF 10", ""])


def long_chain_ligand(n=18, origin=(-8.0, 9.0, 0.0)):
    """A zig-zag chain of n carbons ending in N-H and O-H groups, every C-C bond rotatable: ~22 A long, so its atoms
    span two of model::assign_bonds' 15 A beads (the order of bond lists, and with it bonded_to()'s depth-first
    walk, depends on the bead order)."""
    pts = []
    for i in range(n):
        pts.append((origin[0] + 1.26 * i, origin[1] + (0.45 if i % 2 else -0.45), origin[2] + 0.3 * np.sin(i)))
    atoms = [atom_line(i + 1, f"C{i + 1}", *pts[i], "C") for i in range(n)]
    out = ["ROOT", atoms[0], atoms[1], "ENDROOT"]
    for i in range(2, n):
        out += [f"BRANCH {i:3d} {i + 1:3d}", atoms[i]]
    # polar tail on the last carbon
    last = pts[-1]
    out.append(atom_line(n + 1, "O1", last[0] + 1.2, last[1] + 0.7, last[2], "OA"))
    out.append(atom_line(n + 2, "H1", last[0] + 2.1, last[1] + 0.5, last[2], "HD"))
    for i in range(n - 1, 1, -1):
        out.append(f"ENDBRANCH {i:3d} {i + 1:3d}")
    out += [f"TORSDOF {n - 2}", ""]
    return "\n".join(out)


def box_of(coords, pad=4.0):
    lo, hi = coords.min(0) - pad, coords.max(0) + pad
    return ((lo + hi) / 2).astype(np.float32), (hi - lo).astype(np.float32)


def random_confs(rng, conf0, k, spread=1.0, tors=3.0, small=False):
    """k conformations around conf0: small = gentle moves (stay inside the box, moderate energies)."""
    out = []
    for _ in range(k):
        c = np.array(conf0, dtype=np.float32)
        c[:3] += rng.uniform(-spread, spread, 3).astype(np.float32)
        q = rng.normal(size=4) * (0.15 if small else 1.0)
        if small:
            q[0] += 1
        c[3:7] = (q / np.linalg.norm(q)).astype(np.float32)
        c[7:] = rng.uniform(-tors, tors, len(c) - 7).astype(np.float32) * (0.2 if small else 1.0)
        out.append(c)
    return np.stack(out)


def user_grid_text(center, nelem, spacing, seed=5, amplitude=3.0):
    """A --user_grid file (AutoDock-map style header, main.cpp:635-670): three header lines, SPACING, NELEMENTS, CENTER,
    then (nx+2)(ny+2)(nz+2)-ish value lines -- setup_user_gd turns NELEMENTS n into ceil((n + 1) * g / g) intervals.
    Returns (text, value_lines) -- the latter is what grid::init reads after the header."""
    rng = np.random.RandomState(seed)
    n = [int(np.ceil(np.float32((k + 1) * spacing) / np.float32(spacing))) for k in nelem]
    vals = rng.uniform(-amplitude, amplitude, size=n[0] * n[1] * n[2])
    lines = "\n".join("%.5f" % v for v in vals) + "\n"
    head = ("GRID_PARAMETER_FILE user.gpf\nGRID_DATA_FILE user.fld\nMACROMOLECULE rec.pdbqt\n"
            "SPACING %.3f\nNELEMENTS %d %d %d\nCENTER %.3f %.3f %.3f\n" % ((spacing,) + tuple(nelem) + tuple(center)))
    return head + lines, lines
