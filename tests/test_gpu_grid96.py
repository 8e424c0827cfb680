"""BASELINE config 5: the dynamic-global-pool Dense networks on a 0.25 A / 96^3 grid.
Goldens: tests/golden/cnn_goldens_96.npz = the reference's own dense_1.3*.pt on the oracle's 96^3 grids."""
import os

import numpy as np
import pytest

from oracle import cnn_ref, voxel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")
RES, DIM = 0.25, 23.75


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "cnn_goldens.npz")), np.load(os.path.join(golden_dir, "cnn_goldens_96.npz"))


def atoms(CG, name):
    return tuple(CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))


@pytest.mark.parametrize("name", ["dense_1_3", "dense_1_3_PT_KD_3"])
def test_scores_at_96_match_reference_torchscript(capi, G, name):
    CG, G96 = G
    rec_xyz, rec_smt, lig_smt, poses = atoms(CG, name)
    m = capi.Model(name, resolution=RES, dimension=DIM)
    assert m.grid_points == 96 and abs(m.resolution - RES) < 1e-7
    s = capi.Scorer([m])
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_batch(poses[:2], lig_smt)
    assert np.abs(out["pose"] - G96[name + "/pose"]).max() < 1e-4
    assert np.abs(out["affinity"] - G96[name + "/affinity"]).max() < 1e-4
    assert np.abs(out["loss"] - G96[name + "/loss"]).max() < 1e-4 * np.abs(G96[name + "/loss"]).max()


def test_voxelize_96_bit_exact_support(capi, G):
    CG, G96 = G
    name = "dense_1_3"
    rec_xyz, rec_smt, lig_smt, poses = atoms(CG, name)
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    s = capi.Scorer([capi.Model(name, resolution=RES, dimension=DIM)])
    s.set_receptor(rec_xyz, rec_smt)
    grids, cen = s.voxelize_batch(poses[:1], lig_smt)
    assert grids.shape == (1, 28, 96, 96, 96)
    ref, c = voxel.voxelize_pose(rec_xyz, rec_smt, poses[0], lig_smt, rmap, lmap, None, RES, DIM, blob.radius_scaling)
    assert np.array_equal(c, cen[0])
    assert np.array_equal(ref != 0, grids[0] != 0)
    assert np.abs(ref - grids[0]).max() < 1e-5
    assert int((grids[0] != 0).sum()) == int(G96[name + "/grid_nnz"][0])


def test_gradient_at_96_matches_oracle(capi, G):
    CG, _ = G
    name = "dense_1_3"
    rec_xyz, rec_smt, lig_smt, poses = atoms(CG, name)
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    s = capi.Scorer([capi.Model(name, resolution=RES, dimension=DIM)])
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_grad(poses[:1], lig_smt)
    grid, cen = voxel.voxelize_pose(rec_xyz, rec_smt, poses[0], lig_smt, rmap, lmap, None, RES, DIM,
                                    blob.radius_scaling)
    loss, gg = cnn_ref.loss_and_grid_gradient(blob, grid[None])
    ch, rad = voxel.type_atoms(lig_smt, lmap[0])
    ch = np.where(ch >= 0, ch + rmap[1], -1)
    g0 = voxel.grid_backward(cen, poses[0], ch, rad, rmap[1] + lmap[1], gg[0].numpy(), RES, DIM, blob.radius_scaling)
    assert abs(out["loss"][0] - float(loss[0])) < 1e-3 * max(1.0, abs(float(loss[0])))
    assert np.abs(out["lig_grad"][0] - g0).max() < 2e-3 * np.abs(g0).max()


def test_fixed_head_models_refuse_other_grids(capi):
    for name in ("default2017", "crossdock_default2018", "dense"):
        with pytest.raises(capi.MiGninaError) as ei:
            capi.Model(name, resolution=RES, dimension=DIM)
        assert "96" in str(ei.value)
    # same grid: fine for every family
    assert capi.Model("default2017", resolution=0.5, dimension=23.5).grid_points == 48
