"""Native PDBQT reader (gnina_amd/host/pdbqt.{h,cpp}).  The reference ships no ligand .pdbqt
with expected types / pairs ("parity unpinned"): the expectations below are derived by hand from
parse_pdbqt.cpp / parsing.h / model.cpp for small molecules, plus structural invariants checked with the
torsion-tree oracle (oracle/vina.py: set_conf must reproduce the file's coordinates)."""
import os

import numpy as np
import pytest

from oracle import vina as ovina

REF_DATA = "/root/reference/test/gnina/data"


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import build, capi as c
    build.build()
    return c


def atom_line(serial, name, x, y, z, adtype, q=0.0, het=False):
    rec = "HETATM" if het else "ATOM  "
    return f"{rec}{serial:5d} {name:<4s} LIG A   1    {x:8.3f}{y:8.3f}{z:8.3f}{1.0:6.2f}{0.0:6.2f}    {q:6.3f} {adtype:<2s}"


# zig-zag chain C1-C2-C3-C4-O5-H6, three rotatable bonds
CHAIN = [(0.000, 0.000, 0.0), (1.520, 0.000, 0.0), (2.280, 1.316, 0.0), (3.800, 1.316, 0.0), (4.510, 2.546, 0.0),
         (5.470, 2.546, 0.0)]


def chain_pdbqt():
    a = [atom_line(i + 1, n, *CHAIN[i], t) for i, (n, t) in
         enumerate([("C1", "C"), ("C2", "C"), ("C3", "C"), ("C4", "C"), ("O5", "OA"), ("H6", "HD")])]
    return "\n".join(["REMARK  hand-made test ligand", "ROOT", a[0], a[1], "ENDROOT", "BRANCH   2   3", a[2],
                      "BRANCH   3   4", a[3], "BRANCH   4   5", a[4], a[5], "ENDBRANCH   4   5", "ENDBRANCH   3   4",
                      "ENDBRANCH   2   3", "TORSDOF 3", ""])


def test_ligand_structure_matches_hand_derivation(capi):
    lig = capi.read_pdbqt_ligand(chain_pdbqt(), is_text=True)
    # atom order: a branch's first ("immobile") atom is stored in the parent segment (parsing.h:151-202)
    assert lig["serial"].tolist() == [1, 2, 3, 4, 5, 6]
    assert lig["parent"].tolist() == [-1, 0, 1, 2]
    assert lig["abeg"].tolist() == [0, 3, 4, 5] and lig["aend"].tolist() == [3, 4, 5, 6]
    assert lig["n_tors"] == 3 and lig["torsdof"] == 3
    xyz = np.array(CHAIN, dtype=np.float32)
    np.testing.assert_allclose(lig["coords0"], xyz, atol=1e-6)
    # frames: root origin = first root atom; segment origin = immobile atom, axis from the parent-side atom (tree.h:152-203)
    np.testing.assert_allclose(lig["rel_origin"][1], xyz[2] - xyz[0], atol=1e-6)
    np.testing.assert_allclose(lig["rel_origin"][2], xyz[3] - xyz[2], atol=1e-6)
    np.testing.assert_allclose(lig["rel_origin"][3], xyz[4] - xyz[3], atol=1e-6)
    for k, (a, b) in enumerate([(1, 2), (2, 3), (3, 4)], start=1):
        ax = (xyz[b] - xyz[a]) / np.linalg.norm(xyz[b] - xyz[a])
        np.testing.assert_allclose(lig["rel_axis"][k], ax, atol=1e-6)
    np.testing.assert_allclose(lig["local_xyz"][:3], xyz[:3] - xyz[0], atol=1e-6)
    np.testing.assert_allclose(lig["local_xyz"][3], xyz[3] - xyz[2], atol=1e-6)   # C4 lives in the C3-origin segment
    np.testing.assert_allclose(lig["local_xyz"][4], xyz[4] - xyz[3], atol=1e-6)
    np.testing.assert_allclose(lig["local_xyz"][5], xyz[5] - xyz[4], atol=1e-6)
    # types (adjust_smina_type): C bonded only to C -> hydrophobe 2; C4 bonded to O -> 3; OA with a polar H -> 12
    assert lig["smt"].tolist() == [2, 2, 2, 3, 12, 1]
    # pairs (initialize_pairs): heavy, variable distance, more than 3 bonds apart -> only C1..O5
    assert lig["pairs"].tolist() == [[0, 4]]
    np.testing.assert_allclose(lig["conf0"], [0, 0, 0, 1, 0, 0, 0, 0, 0, 0], atol=1e-7)


def test_tree_reproduces_the_file_and_torsions_preserve_bonds(capi):
    lig = capi.read_pdbqt_ligand(chain_pdbqt(), is_text=True)
    h = ovina.LigandHandle(lig)
    coords, _, _ = ovina.set_conf(h, lig["conf0"])
    assert np.abs(coords - lig["coords0"]).max() < 1e-5
    rng = np.random.RandomState(0)
    conf = lig["conf0"].copy()
    conf[7:] = rng.uniform(-3, 3, 3)
    moved, _, _ = ovina.set_conf(h, conf)
    bonds = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]
    for a, b in bonds:
        d0 = np.linalg.norm(lig["coords0"][a] - lig["coords0"][b])
        assert abs(np.linalg.norm(moved[a] - moved[b]) - d0) < 1e-4
    # bond angles survive too (rotation about the bond axis only); end-to-end distance changes
    for a, b, c in [(0, 1, 2), (1, 2, 3), (2, 3, 4), (3, 4, 5)]:
        d0 = np.linalg.norm(lig["coords0"][a] - lig["coords0"][c])
        assert abs(np.linalg.norm(moved[a] - moved[c]) - d0) < 1e-4
    assert abs(np.linalg.norm(moved[0] - moved[4]) - np.linalg.norm(lig["coords0"][0] - lig["coords0"][4])) > 1e-3
    assert np.abs(moved[:3] - lig["coords0"][:3]).max() < 1e-5            # the root did not move


def test_single_atom_branch_makes_no_torsion_and_typing_rules(capi):
    # methylamine-like: N (root) - C, the C as a one-atom branch: "essentially empty" -> no segment (parsing.h:204-211)
    lines = ["ROOT", atom_line(1, "N1", 0, 0, 0, "N"), atom_line(2, "H1", 0.0, 0.95, 0.3, "HD"), "ENDROOT",
             "BRANCH   1   3", atom_line(3, "C1", 1.47, 0, 0, "C"), "ENDBRANCH   1   3", "TORSDOF 0", ""]
    lig = capi.read_pdbqt_ligand("\n".join(lines), is_text=True)
    assert lig["n_tors"] == 0 and lig["serial"].tolist() == [1, 3, 2]     # the branch atom follows its parent atom
    # N with a polar hydrogen -> NitrogenXSDonor (7); the carbon next to N -> non-hydrophobe (3)
    assert lig["smt"].tolist() == [7, 3, 1]
    assert lig["pairs"].shape == (0, 2)


def test_receptor_typing_and_reference_file(capi, tmp_path):
    # water (OA + 2 HD) -> donor/acceptor 12; a lone carbonyl-like OA -> acceptor 13; NA without H -> 9; N with H -> 7;
    # an aromatic carbon next to N -> 5, far from everything -> 4; zinc and an unknown two-letter metal -> 23 / 26
    atoms = [(1, "O", 0, 0, 0, "OA"), (2, "H1", 0.96, 0, 0, "HD"), (3, "H2", -0.24, 0.93, 0, "HD"),
             (4, "O2", 10, 0, 0, "OA"), (5, "N1", 20, 0, 0, "NA"), (6, "N2", 30, 0, 0, "N"), (7, "H3", 30.9, 0.4, 0, "HD"),
             (8, "CA", 31.0 - 2.3, 0.0, 0.9, "A"), (9, "CB", 50, 0, 0, "A"), (10, "ZN", 60, 0, 0, "Zn"),
             (11, "CU", 70, 0, 0, "Cu")]
    atoms[7] = (8, "CA", 30.0, -1.35, 0.0, "A")
    text = "\n".join(["REMARK receptor"] + [atom_line(*a) for a in atoms] + ["TER", "END", ""])
    p = tmp_path / "rec.pdbqt"
    p.write_text(text)
    xyz, smt = capi.read_pdbqt_receptor(str(p))
    assert len(smt) == 11 and np.allclose(xyz[3], [10, 0, 0])
    assert smt.tolist() == [12, 1, 1, 13, 9, 7, 1, 5, 4, 23, 26]
    ref = os.path.join(REF_DATA, "GSK3B_DFG_out_35-388-processed_rigid.pdbqt")
    if os.path.exists(ref):   # the reference's own rigid-receptor fixture (only in the build container)
        xyz, smt = capi.read_pdbqt_receptor(ref)
        n_lines = sum(1 for l in open(ref) if l.startswith("ATOM  ") or l.startswith("HETATM"))
        assert len(smt) == n_lines > 1000 and ((smt >= 0) & (smt < 28)).all()
        # a protein has backbone N-H donors, carbonyl acceptors, both kinds of carbon and polar hydrogens
        for t in (1, 2, 3, 7, 13):
            assert (smt == t).sum() > 20, t
        assert (smt == 0).sum() == 0          # PDBQT receptors carry polar hydrogens only
        # every polar hydrogen sits on a donor: count of donors (N 7/8, O 11/12) is between #HD/3 and #HD
        donors = np.isin(smt, [7, 8, 11, 12]).sum()
        assert (smt == 1).sum() / 3 <= donors <= (smt == 1).sum()


def test_errors_carry_file_and_line(capi, tmp_path):
    bad = chain_pdbqt().replace("ENDBRANCH   3   4", "ENDBRANCH   3   9")
    with pytest.raises(capi.MiGninaError, match=r"<text>:\d+: Inconsistent branch numbers"):
        capi.read_pdbqt_ligand(bad, is_text=True)
    with pytest.raises(capi.MiGninaError, match="Missing TORSDOF"):
        capi.read_pdbqt_ligand(chain_pdbqt().replace("TORSDOF 3", ""), is_text=True)
    with pytest.raises(capi.MiGninaError, match="not a valid AutoDock type|GenericMetal|valid"):
        capi.read_pdbqt_ligand(chain_pdbqt().replace(" OA", " Qx7"), is_text=True)
    with pytest.raises(capi.MiGninaError, match="No atom number 7"):
        capi.read_pdbqt_ligand(chain_pdbqt().replace("BRANCH   2   3", "BRANCH   7   3", 1), is_text=True)
    with pytest.raises(capi.MiGninaError, match="could not open"):
        capi.read_pdbqt_receptor(str(tmp_path / "none.pdbqt"))
    with pytest.raises(capi.MiGninaError, match="MODEL"):
        capi.read_pdbqt_ligand("MODEL 1\n" + chain_pdbqt(), is_text=True)


def test_pose_writer_round_trip(capi):
    """gnina's .pdbqt output (result_info.cpp:151-164, model.cpp:779-810): the written poses parse back to the
    written coordinates, remarks follow the reference's rules, non-ATOM lines survive untouched."""
    text = chain_pdbqt()
    lig = capi.read_pdbqt_ligand(text, is_text=True)
    h = ovina.LigandHandle(lig)
    rng = np.random.RandomState(1)
    poses = []
    for _ in range(3):
        conf = lig["conf0"].copy()
        conf[:3] += rng.uniform(-5, 5, 3)
        conf[7:] = rng.uniform(-3, 3, 3)
        poses.append(ovina.set_conf(h, conf)[0])
    poses = np.stack(poses)
    out = capi.pdbqt_poses_text(text, poses, energies=[-7.25, -6.5, 1e3], cnnscores=[0.91, 0.5, -1.0],
                                cnnaffinities=[6.75, 0.0, 5.5], is_text=True)
    models = out.split("ENDMDL\n")[:-1]
    assert len(models) == 3 and models[0].startswith("MODEL 1\nREMARK minimizedAffinity -7.25\nREMARK CNNscore 0.910000026\nREMARK CNNaffinity 6.75\n")
    assert "CNNaffinity" not in models[1] and "REMARK CNNscore 0.5\n" in models[1]       # affinity 0 -> omitted
    assert "CNNscore" not in models[2] and "REMARK CNNaffinity 5.5\n" in models[2]       # score < 0 -> omitted
    for k, m in enumerate(models):
        body = "\n".join(l for l in m.split("\n") if not l.startswith(("MODEL", "REMARK minimized", "REMARK CNN")))
        back = capi.read_pdbqt_ligand(body, is_text=True)
        assert np.abs(back["coords0"] - np.round(poses[k].astype(np.float64), 3)).max() < 2e-3
        assert back["smt"].tolist() == lig["smt"].tolist() and back["pairs"].tolist() == lig["pairs"].tolist()
        assert "REMARK  hand-made test ligand" in m and "TORSDOF 3" in m and "BRANCH   3   4" in m
    with pytest.raises(capi.MiGninaError, match="8-column"):
        capi.pdbqt_poses_text(text, poses * 1e5, energies=[0, 0, 0], is_text=True)


SER_FLEX = "\n".join([
    "BEGIN_RES SER A  10", "REMARK  2 active torsions", "ROOT", atom_line(1, "CA", 0.0, 0.0, 0.0, "C"), "ENDROOT",
    "BRANCH   1   2", atom_line(2, "CB", 1.52, 0.0, 0.0, "C"),
    "BRANCH   2   3", atom_line(3, "OG", 2.10, 1.30, 0.0, "OA"), atom_line(4, "HG", 3.05, 1.30, 0.0, "HD"),
    "ENDBRANCH   2   3", "ENDBRANCH   1   2", "END_RES", ""])
RIGID = "\n".join([atom_line(10, "N", -0.53, 1.36, 0.0, "N"), atom_line(11, "CX", 2.10, 2.80, 0.0, "C"), "TER", ""])


def test_flexible_receptor_rows_and_typing(capi, tmp_path):
    """parse_receptor_pdbqt(rigid, flex): postprocess_residue makes the ROOT atom and the first atom of the top-level
    branch inflex ([CA, CB]), the rest movable in tree order ([OG, HG]); rows come back movable | inflex | rigid
    (DLScorer::setReceptor).  Typing on the combined model: OG carries HG -> donor/acceptor 12; CB is bonded to OG
    -> 3; CA is bonded (inflex-rigid distances are fixed) to the backbone N of the rigid part -> 3; the rigid carbon
    1.5 A from OG is NOT bonded to it (rigid-movable distances are variable, model.cpp:491-508) -> stays 2."""
    xyz, smt, nm, ni = capi.read_pdbqt_receptor_flex(RIGID, SER_FLEX, is_text=True)
    assert (nm, ni, len(smt)) == (2, 2, 6)
    assert np.allclose(xyz, [[2.10, 1.30, 0], [3.05, 1.30, 0], [0, 0, 0], [1.52, 0, 0], [-0.53, 1.36, 0], [2.10, 2.80, 0]])
    assert smt.tolist() == [12, 1, 3, 3, 6, 2]
    # the same six atoms as one rigid receptor: now the carbon next to OG is bonded to a heteroatom
    allrigid = "\n".join(l for l in (SER_FLEX + RIGID).split("\n") if l.startswith("ATOM"))
    p = tmp_path / "all.pdbqt"
    p.write_text(allrigid + "\n")
    _, smt_r = capi.read_pdbqt_receptor(str(p))
    assert smt_r.tolist() == [3, 3, 12, 1, 6, 3]
    # files on disk, two residues: the second residue's atoms follow the first's inside each group
    f, r = tmp_path / "flex.pdbqt", tmp_path / "rigid.pdbqt"
    second = SER_FLEX.replace("SER A  10", "SER A  11")
    for old, new in (("   0.000   0.000   0.000", "  20.000   0.000   0.000"), ("   1.520   0.000", "  21.520   0.000"),
                     ("   2.100   1.300", "  22.100   1.300"), ("   3.050   1.300", "  23.050   1.300")):
        second = second.replace(old, new)
    f.write_text(SER_FLEX + second)
    r.write_text(RIGID)
    xyz2, smt2, nm2, ni2 = capi.read_pdbqt_receptor_flex(str(r), str(f))
    assert (nm2, ni2, len(smt2)) == (4, 4, 10)
    assert np.allclose(xyz2[:4, 0], [2.10, 3.05, 22.10, 23.05]) and np.allclose(xyz2[4:8, 0], [0, 1.52, 20, 21.52])
    assert smt2.tolist() == [12, 1, 12, 1, 3, 3, 2, 3, 6, 2]     # the second CA has no rigid N next to it -> 2
    with pytest.raises(capi.MiGninaError, match=r"<flex>:\d+: Unknown or inappropriate tag"):
        capi.read_pdbqt_receptor_flex(RIGID, SER_FLEX.replace("END_RES", "ENDRES"), is_text=True)
    with pytest.raises(capi.MiGninaError, match="Unknown or inappropriate tag"):
        capi.read_pdbqt_receptor_flex(RIGID, "ROOT\n" + SER_FLEX, is_text=True)


def test_pdbqt_to_gninatypes_tool(capi, tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r, f, out = tmp_path / "r.pdbqt", tmp_path / "f.pdbqt", tmp_path / "o.gninatypes"
    r.write_text(RIGID)
    f.write_text(SER_FLEX)
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "pdbqt_to_gninatypes.py"), "--flex", str(f), str(r),
                          str(out)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "2 movable + 2 inflex + 2 rigid" in res.stdout, res.stderr
    xyz, smt = capi.read_gninatypes(str(out))
    assert smt.tolist() == [12, 1, 3, 3, 6, 2] and np.allclose(xyz[0], [2.10, 1.30, 0.0])


def test_sdf_pose_output_follows_result_info_write(capi):
    """gnina's native .sdf output (result_info.cpp:117-160 + sdfcontext::write, model.cpp:827-907), reconstructed line
    by line: counts line, %10.4f coordinates, 3-wide bond fields, charge property, tags with 5 / 10 decimals, CNN_VS =
    affinity x score, optional tags left out, $$$$."""
    el = ["C", "O", "N", "Cl"]
    xyz = np.array([[0, 0, 0], [1.25, 0.5, -0.125], [2.5, 0, 0], [-10.33337, 123.456789, 0.00004]], dtype=np.float32)
    bonds = [(0, 1, 2), (1, 2, 1), (0, 3, 1)]
    text = capi.sdf_pose_text("lig one", el, xyz, bonds, -7.123456, rmsd=1.5, cnnscore=0.987654321, cnnaffinity=6.5,
                              cnnvariance=0.25, props=[("c", 2, 1)])
    want = "\n".join([
        "lig one", "", "",
        "  4  3  0  0  0  0  0  0  0  0999 V2000",
        "    0.0000    0.0000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0",
        "    1.2500    0.5000   -0.1250 O   0  0  0  0  0  0  0  0  0  0  0  0",
        "    2.5000    0.0000    0.0000 N   0  0  0  0  0  0  0  0  0  0  0  0",
        "  -10.3334  123.4568    0.0000 Cl  0  0  0  0  0  0  0  0  0  0  0  0",
        "  1  2  2  0", "  2  3  1  0", "  1  4  1  0",
        "M  CHG 1   3   1",
        "M  END",
        "> <minimizedAffinity>", "-7.12346", "",
        "> <minimizedRMSD>", "1.50000", "",
        "> <CNNscore>", "0.9876543283", "",
        "> <CNNaffinity>", "6.5000000000", "",
        "> <CNN_VS>", "%.10f" % float(np.float32(6.5) * np.float32(0.987654321)), "",
        "> <CNNaffinity_variance>", "0.2500000000", "",
        "$$$$", ""])
    assert text == want
    plain = capi.sdf_pose_text("x", el[:2], xyz[:2], bonds[:1], -1.0)      # rescoring off: no CNN tags, no rmsd
    assert "> <CNNscore>" not in plain and "> <minimizedRMSD>" not in plain and "> <CNNaffinity>" not in plain
    assert plain.endswith("> <minimizedAffinity>\n-1.00000\n\n$$$$\n")
    # atom_index: SDF atom i takes the coordinates of model atom atom_index[i]
    perm = capi.sdf_pose_text("x", el[:2], xyz, bonds[:1], 0.0, atom_index=[2, 1])
    assert "    2.5000    0.0000    0.0000 C  " in perm
