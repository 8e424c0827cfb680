"""The Overlap toy models (test/gnina/data/overlap.pt, overlap_smallr.pt) and the reference's own behavioural test of
the CNN gradient path, test/gnina/test_min.py: minimising the CNN loss of the overlap model must move the ligand
atoms onto the receptor atoms (within 0.1 A, `are_similar`).  Also exercises skip_softmax / apply_logistic_loss
(torch_model.cpp:188-195).  Goldens: the reference .pt files on the oracle's grids (make_overlap_goldens.py)."""
import os

import numpy as np
import pytest

from oracle import cnn_ref, voxel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")
C = 2


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def OG(golden_dir):
    return np.load(os.path.join(golden_dir, "overlap_goldens.npz"))


@pytest.mark.parametrize("name", ["overlap", "overlap_smallr"])
def test_scores_and_gradients_match_reference_module(capi, OG, name):
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    for case in ("C_C1", "CC_CC2", "rand8", "far"):
        k = f"{name}/{case}/"
        rec, lig = OG[k + "rec"], OG[k + "lig"]
        rs, ls = np.full(len(rec), C, np.int32), np.full(len(lig), C, np.int32)
        s = capi.Scorer([name])
        s.set_receptor(rec, rs)
        out = s.score_batch(lig[None], ls)
        assert abs(out["pose"][0] - OG[k + "pose"]) <= 2e-5 * OG[k + "pose"] + 1e-30, (case, out["pose"][0])
        assert abs(out["loss"][0] - OG[k + "loss"]) < 1e-4 and out["affinity"][0] == 0.0
        g = s.score_grad(lig[None], ls)
        assert abs(g["loss"][0] - OG[k + "loss"]) < 1e-4
        grid, cen = voxel.voxelize_pose(rec, rs, lig, ls, rmap, lmap, None, blob.resolution, blob.dimension,
                                        blob.radius_scaling)
        loss, gg = cnn_ref.loss_and_grid_gradient(blob, grid[None])
        ch, rad = voxel.type_atoms(ls, lmap[0])
        g0 = voxel.grid_backward(cen, lig, np.where(ch >= 0, ch + rmap[1], -1), rad, 2, gg[0].numpy(), blob.resolution,
                                 blob.dimension, blob.radius_scaling)
        if case == "far":
            assert not g["lig_grad"].any() and not g0.any()      # the where(ave > 0, ...) branch has no gradient
        else:
            assert np.abs(g["lig_grad"][0] - g0).max() < 2e-3 * np.abs(g0).max(), case


def one_node_ligand(coords):
    """a rigid ligand (no torsions) in the mi_ligand_desc layout"""
    coords = np.asarray(coords, dtype=np.float32)
    n = len(coords)
    return {"smt": np.full(n, C, np.int32), "local_xyz": (coords - coords[0]).astype(np.float32),
            "parent": np.array([-1], np.int32), "abeg": np.array([0], np.int32), "aend": np.array([n], np.int32),
            "rel_origin": np.zeros((1, 3), np.float32), "rel_axis": np.zeros((1, 3), np.float32),
            "pairs": np.zeros((0, 2), np.int32), "n_tors": 0,
            "conf0": np.concatenate([coords[0], [1, 0, 0, 0]]).astype(np.float32)}


def are_similar(a, b):      # test_min.py:21-38: a bijection of atoms closer than 0.1 A
    a, b = np.asarray(a), np.asarray(b)
    used = set()
    for x in a:
        d = np.linalg.norm(b - x, axis=1)
        j = [i for i in np.argsort(d) if i not in used and d[i] < 0.1]
        if not j:
            return False
        used.add(j[0])
    return True


@pytest.mark.parametrize("rec,lig", [([[0, 0, 0]], [[1, 1, 1]]),                                     # C.xyz / C1.xyz
                                     ([[0, 0, 0], [1.6, 0, 0]], [[1.0, -3.2, 1.0], [1.0, -1.6, 1.0]])])  # CC.xyz / CC2.xyz
def test_min_overlap_converges_onto_the_receptor(capi, rec, lig):
    """test_min.py:64-84: `--cnn_scoring refinement --cnn_model overlap.pt --minimize` -> output similar to the receptor."""
    rec = np.asarray(rec, np.float32)
    s = capi.Scorer(["overlap"])
    s.set_receptor(rec, np.full(len(rec), C, np.int32))
    desc = one_node_ligand(lig)
    v = capi.Vina()
    v.set_ligand(desc)
    box = capi.CnnBox.make(23.5)                    # no search box, the CNN cube only
    start = desc["conf0"][None]
    assert not are_similar(rec, v.coords_batch(start)[0])
    e0, _ = v.cnn_eval_batch(s, start, box, None, deriv=False)
    e, out, tries, evals = v.cnn_refine_batch(s, start, box, max_iters=10000)   # main.cpp:1157-1158
    final = v.coords_batch(out)[0]
    assert e[0] < e0[0] - 0.3 and tries[0] == 1
    assert are_similar(rec, final), (final, e, evals)


C8FLAT = [[-4.41300, -0.17760, 0.43050], [-3.15750, -0.17690, -0.43680], [-1.89220, -0.05930, 0.42040],
          [-0.63310, -0.05880, -0.45450], [0.63310, 0.05880, 0.40230], [1.89220, 0.05930, -0.47270],
          [3.15750, 0.17690, 0.38460], [4.41300, 0.17760, -0.48280]]                 # test/gnina/data/C8flat.xyz
C8BENT = [[-3.5067, 1.1289, -0.0582], [-2.5909, -0.0810, -0.2192], [-1.3522, 0.0393, 0.6810], [-0.4171, -1.1759, 0.5481],
          [0.4521, -1.1392, -0.7293], [1.9062, -0.7109, -0.4693], [2.0237, 0.7561, -0.0346], [3.4849, 1.1827, 0.0726]]  # C8bent.sdf


def test_min_fully_flexible_octane(capi):
    """test_min.py:87-104: `-r C8flat.xyz -l C8bent.sdf --cnn_scoring all --cnn_model overlap_smallr.pt --minimize`:
    the bent octane must unfold onto the flat one.  The torsion tree (5 rotatable C-C bonds, root = C4) is written
    as PDBQT and read back by the native reader, so reader, tree kernels and CNN refinement are exercised together."""
    from tests.test_pdbqt_cpu import atom_line
    a = [atom_line(i + 1, f"C{i + 1}", *C8BENT[i], "C") for i in range(8)]
    text = "\n".join(["ROOT", a[3], "ENDROOT",
                      "BRANCH   4   5", a[4], "BRANCH   5   6", a[5], "BRANCH   6   7", a[6], a[7],
                      "ENDBRANCH   6   7", "ENDBRANCH   5   6", "ENDBRANCH   4   5",
                      "BRANCH   4   3", a[2], "BRANCH   3   2", a[1], a[0], "ENDBRANCH   3   2", "ENDBRANCH   4   3",
                      "TORSDOF 5", ""])
    lig = capi.read_pdbqt_ligand(text, is_text=True)
    assert lig["n_tors"] == 5 and (lig["smt"] == 2).all()
    rec = np.asarray(C8FLAT, np.float32)
    s = capi.Scorer(["overlap_smallr"])
    s.set_receptor(rec, np.full(8, C, np.int32))
    v = capi.Vina()
    v.set_ligand(lig)
    box = capi.CnnBox.make(23.5)
    start = lig["conf0"][None]
    assert not are_similar(rec, v.coords_batch(start)[0])
    e0, _ = v.cnn_eval_batch(s, start, box, None, deriv=False)
    e, out, tries, evals = v.cnn_refine_batch(s, start, box, max_iters=10000)
    final = v.coords_batch(out)[0]
    assert e[0] < e0[0] and are_similar(rec, final), (e0, e, evals, final)


@pytest.mark.parametrize("weight", [1.0, 0.7])
def test_min_cnn_plus_empirical_energy_identity(capi, weight):
    """test_min.py:112-140: with --cnn_mix_emp_force --cnn_mix_emp_energy --cnn_empirical_weight w the minimised
    octane does NOT land on the receptor (the empirical term repels), and the reported total energy obeys
    total = (-log(CNNscore) + w * empirical) / (1 + w)   (validate_energies, tolerance 1e-3)."""
    from tests.test_pdbqt_cpu import atom_line
    a = [atom_line(i + 1, f"C{i + 1}", *C8BENT[i], "C") for i in range(8)]
    text = "\n".join(["ROOT", a[3], "ENDROOT", "BRANCH   4   5", a[4], "BRANCH   5   6", a[5], "BRANCH   6   7", a[6], a[7],
                      "ENDBRANCH   6   7", "ENDBRANCH   5   6", "ENDBRANCH   4   5", "BRANCH   4   3", a[2],
                      "BRANCH   3   2", a[1], a[0], "ENDBRANCH   3   2", "ENDBRANCH   4   3", "TORSDOF 5", ""])
    lig = capi.read_pdbqt_ligand(text, is_text=True)
    rec = np.asarray(C8FLAT, np.float32)
    rs = np.full(8, C, np.int32)
    s = capi.Scorer(["overlap_smallr"])
    s.set_receptor(rec, rs)
    v = capi.Vina()
    v.set_receptor(rec, rs)
    lo, hi = rec.min(0) - 30.0, rec.max(0) + 30.0   # wide enough to hold wherever the repulsion sends the ligand
    n = np.ceil((hi - lo) / 0.375).astype(np.int32)
    v.build_cache(lo, lo + 0.375 * n, n, [C], 1e3)       # only so that the direct (non_cache) evaluation has a box
    v.set_ligand(lig)
    box = capi.CnnBox.make(23.5, mix_emp_force=True, mix_emp_energy=True, empirical_weight=weight, v=1000.0)
    start = lig["conf0"][None]
    e, out, tries, evals = v.cnn_refine_batch(s, start, box, max_iters=10000)
    final = v.coords_batch(out)[0]
    assert not are_similar(rec, final)
    total, _ = v.cnn_eval_batch(s, out, box, None, deriv=True)           # "Total energy after refinement"
    score = s.score_batch(final[None], lig["smt"])                       # CNNscore of the output pose
    emp, _, _ = v.eval_batch(out, (1000.0, 1000.0, 1000.0), deriv=False, grid_only=True, direct=True)  # nc_new.eval
    calc = (-np.log(score["pose"][0]) + weight * emp[0]) / (1 + weight)
    # As in the reference (main.cpp:163-167) the total comes from eval_deriv (interpolated pair tables) and the
    # printed empirical energy from eval (midpoint tables): the identity holds up to the table discretisation of
    # the empirical term (a few 0.1 % where the repulsion is steep) on top of test_min.py's 1e-3.
    tol = 1e-3 + 5e-3 * weight * abs(emp[0]) / (1 + weight)
    assert abs(total[0] - calc) < tol, (total, calc, emp, score["pose"])
    assert abs(e[0] - total[0]) < 1e-4 * max(1.0, abs(total[0]))
