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
