"""Rows a9/a10: the gradient path (refinement).  HIP: mi_scorer_score_grad = forward + transposed
convolutions + fused un-pooling + GridMaker::backward.  Oracle: autograd through oracle/cnn_ref.py on
the oracle grid (what loss.backward() does in torch_model.cpp:197-199), then ora_grid_backward
(SURVEY App. A.4).  The oracle itself is checked against finite differences of its own loss."""
import os

import numpy as np
import pytest
import torch

from oracle import cnn_ref, voxel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def CG(golden_dir):
    return np.load(os.path.join(golden_dir, "cnn_goldens.npz"))


def oracle_lig_gradient(blob, rec_xyz, rec_smt, pose, lig_smt):
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    grid, cen = voxel.voxelize_pose(rec_xyz, rec_smt, pose, lig_smt, rmap, lmap)
    loss, gg = cnn_ref.loss_and_grid_gradient(blob, grid[None])
    ch, rad = voxel.type_atoms(lig_smt, lmap[0])
    ch = np.where(ch >= 0, ch + rmap[1], -1)
    g = voxel.grid_backward(cen, pose, ch, rad, rmap[1] + lmap[1], gg[0].numpy(), blob.resolution, blob.dimension,
                            blob.radius_scaling)
    return float(loss[0]), g


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "crossdock_default2018_KD_4", "dense",
                                  "dense_1_3", "dense_1_3_PT_KD_3"])
def test_ligand_gradient_matches_oracle(capi, CG, name):
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_grad(poses, lig_smt)
    fwd = s.score_batch(poses, lig_smt)
    assert np.abs(out["pose"] - fwd["pose"]).max() < 1e-6 and np.abs(out["loss"] - fwd["loss"]).max() < 1e-5
    assert np.abs(out["pose"] - CG[name + "/pose"]).max() < 1e-4
    for b in range(len(poses)):
        loss0, g0 = oracle_lig_gradient(blob, rec_xyz, rec_smt, poses[b], lig_smt)
        scale = max(np.abs(g0).max(), 1e-6)
        assert abs(out["loss"][b] - loss0) < 1e-3 * max(1.0, abs(loss0))
        assert np.abs(out["lig_grad"][b] - g0).max() < 2e-3 * scale, (b, np.abs(out["lig_grad"][b] - g0).max(), scale)


def test_gradient_with_hydrogens_and_ensemble(capi, CG):
    names = ["default2017", "crossdock_default2018"]
    base = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    lig_smt = lig_smt.copy()
    lig_smt[[1, 5]] = 1          # polar hydrogens: untyped rows, gradient must be exactly zero
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_grad(poses[:2], lig_smt)
    assert not out["lig_grad"][:, [1, 5]].any()
    acc = np.zeros((2, len(lig_smt), 3))
    for n in names:
        blob = cnn_ref.Blob(os.path.join(WEIGHTS, n + ".mgw"))
        for b in range(2):
            acc[b] += oracle_lig_gradient(blob, rec_xyz, rec_smt, poses[b], lig_smt)[1] / len(names)
    assert np.abs(out["lig_grad"] - acc).max() < 2e-3 * np.abs(acc).max()


def test_gradient_descends_the_loss(capi, CG):
    """A small step against the gradient must lower the CNN loss (what refinement relies on,
    test/gnina/test_cnn.py:56-59)."""
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_grad(poses, lig_smt)
    g = out["lig_grad"]
    step = 0.02 / max(np.abs(g).max(), 1e-9)
    out2 = s.score_batch(poses - step * g, lig_smt)
    assert (out2["loss"] < out["loss"] + 1e-7).all() and (out2["loss"] < out["loss"]).any()


def test_default_ensemble_gradient(capi, CG):
    """gnina's default ensemble (cnn_torch_scorer.cpp:33-35): two Dense models + one Default2018 model; the
    ligand gradient is the ensemble mean (m.scale_minus_forces(1 / cnt), cnn_torch_scorer.cpp:172-175)."""
    names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
    base = "dense_1_3"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_grad(poses[:2], lig_smt)
    acc = np.zeros((2, len(lig_smt), 3))
    for n in names:
        blob = cnn_ref.Blob(os.path.join(WEIGHTS, n + ".mgw"))
        for b in range(2):
            acc[b] += oracle_lig_gradient(blob, rec_xyz, rec_smt, poses[b], lig_smt)[1] / len(names)
    assert np.abs(out["lig_grad"] - acc).max() < 2e-3 * np.abs(acc).max()


@pytest.mark.parametrize("name", ["crossdock_default2018", "dense_1_3"])
def test_flexible_receptor_rows(capi, CG, name):
    """flexible-residue atoms get per-pose coordinates (dl_scorer.cpp:181-192) and their own
    gradient (getReceptorGradient, cnn_torch_scorer.cpp:216-224)."""
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    poses = poses[:2]
    cen = poses.mean(axis=1)
    near = np.argsort(np.linalg.norm(rec_xyz - cen[0], axis=1))[:14]
    rows = np.sort(near)[::-1].copy()            # deliberately not ascending
    rng = np.random.default_rng(5)
    flex = rec_xyz[rows][None] + rng.normal(0, 0.3, (2, len(rows), 3)).astype(np.float32)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    s.set_flex(rows)
    out = s.score_flex(poses, lig_smt, flex)
    fwd = s.score_flex(poses, lig_smt, flex, grad=False)
    assert np.abs(out["pose"] - fwd["pose"]).max() < 1e-6
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    for b in range(2):
        rx = rec_xyz.copy()
        rx[rows] = flex[b]
        # the same pose through a scorer whose receptor simply has the moved coordinates
        s2 = capi.Scorer([name])
        s2.set_receptor(rx, rec_smt)
        ref = s2.score_batch(poses[b:b + 1], lig_smt)
        assert out["pose"][b] == ref["pose"][0] and out["affinity"][b] == ref["affinity"][0]
        grid, c = voxel.voxelize_pose(rx, rec_smt, poses[b], lig_smt, rmap, lmap)
        loss, gg = cnn_ref.loss_and_grid_gradient(blob, grid[None])
        ch, rad = voxel.type_atoms(rec_smt[rows], rmap[0])
        g0 = voxel.grid_backward(c, flex[b], ch, rad, rmap[1] + lmap[1], gg[0].numpy(), blob.resolution,
                                 blob.dimension, blob.radius_scaling)
        scale = max(np.abs(g0).max(), 1e-6)
        assert np.abs(out["flex_grad"][b] - g0).max() < 2e-3 * scale
        _, gl = oracle_lig_gradient(blob, rx, rec_smt, poses[b], lig_smt)
        assert np.abs(out["lig_grad"][b] - gl).max() < 2e-3 * max(np.abs(gl).max(), 1e-6)
    # rows that do not move keep working after the declaration is cleared
    s.set_flex(np.zeros(0, np.int32))
    plain = s.score_batch(poses, lig_smt)
    assert np.abs(plain["pose"] - CG[name + "/pose"][:2]).max() < 1e-4


def test_flexible_receptor_from_pdbqt_files(capi):
    """The rows of mi_pdbqt_read_receptor_flex go straight into mi_scorer_set_receptor / mi_scorer_set_flex (movable
    side-chain atoms first): scoring with moved side-chain coordinates equals scoring a receptor whose rows were
    moved, and the side-chain gradient is non-zero where the ligand touches it."""
    from tests.test_pdbqt_cpu import RIGID, SER_FLEX
    rec_xyz, rec_smt, n_mov, n_inflex = capi.read_pdbqt_receptor_flex(RIGID, SER_FLEX, is_text=True)
    assert (n_mov, n_inflex) == (2, 2)
    lig = np.array([[[3.5, 2.6, 1.0], [4.6, 3.4, 1.2], [4.4, 1.4, 0.4]]], dtype=np.float32)
    lig_smt = np.array([2, 13, 6], dtype=np.int32)
    moved = rec_xyz[:n_mov][None] + np.array([[[0.4, -0.3, 0.2], [0.5, -0.2, 0.3]]], dtype=np.float32)
    s = capi.Scorer(["default2017"])
    s.set_receptor(rec_xyz, rec_smt)
    s.set_flex(np.arange(n_mov, dtype=np.int32))
    out = s.score_flex(lig, lig_smt, moved)
    ref_xyz = rec_xyz.copy()
    ref_xyz[:n_mov] = moved[0]
    s2 = capi.Scorer(["default2017"])
    s2.set_receptor(ref_xyz, rec_smt)
    ref = s2.score_batch(lig, lig_smt)
    assert out["pose"][0] == ref["pose"][0] and out["affinity"][0] == ref["affinity"][0]
    assert np.abs(out["flex_grad"][0][0]).max() > 0 and not out["flex_grad"][0][1].any()   # OG typed, HG (polar H) not
