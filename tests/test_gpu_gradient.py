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
    # (Default2017 / Default2018: the gradient call's forward half computes the forward program's bits.  Dense: the forward
    # program folds the blocks' BatchNorm into the split weights (conv3d_h2_dense.hip), the gradient program applies it while
    # staging -- same arithmetic to ~1e-6, not the same bits)
    tol = 5e-6 if name.startswith("dense") else 1e-6
    assert np.abs(out["pose"] - fwd["pose"]).max() < tol and np.abs(out["loss"] - fwd["loss"]).max() < 1e-5 * max(1.0, float(np.abs(fwd["loss"]).max()))
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
    assert np.abs(out["pose"] - fwd["pose"]).max() < (5e-6 if name.startswith("dense") else 1e-6)
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    for b in range(2):
        rx = rec_xyz.copy()
        rx[rows] = flex[b]
        # the same pose through a scorer whose receptor simply has the moved coordinates
        s2 = capi.Scorer([name])
        s2.set_receptor(rx, rec_smt)
        ref = s2.score_batch(poses[b:b + 1], lig_smt)
        if name.startswith("dense"):   # (gradient program vs forward program: see test_ligand_gradient_matches_oracle)
            assert abs(out["pose"][b] - ref["pose"][0]) < 5e-6 and abs(out["affinity"][b] - ref["affinity"][0]) < 5e-5
            assert fwd["pose"][b] == ref["pose"][0] and fwd["affinity"][b] == ref["affinity"][0]   # forward vs forward: bits
        else:
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


def _rescaled_model(name, i, k, tmp_path):
    """`name` with convolution i (weights and bias) times k and the weights of the layer behind it -- the next convolution,
    or the fully connected heads -- times 1 / k (k a power of two): ReLU and both pools commute with a positive factor, so
    every score keeps its bits, the activations of that one layer are k times and their gradients 1 / k times the shipped
    model's"""
    import struct
    raw = bytearray(open(os.path.join(WEIGHTS, name + ".mgw"), "rb").read())
    blob = cnn_ref.Blob(bytes(raw))
    (hl,) = struct.unpack("<I", raw[8:12])
    off = 12 + hl
    off += (-off) % 64
    data = np.frombuffer(raw, dtype="<f4", offset=off).copy()
    convs = [t for t in blob.ops if t[0] == "conv"]
    i %= len(convs)
    kk, cin, cout, w_off, b_off = int(convs[i][1]), int(convs[i][4]), int(convs[i][5]), int(convs[i][8]), int(convs[i][9])
    data[w_off:w_off + kk ** 3 * cin * cout] *= np.float32(k)
    data[b_off:b_off + cout] *= np.float32(k)
    if i + 1 < len(convs):
        kk, cin, cout, w_off = int(convs[i + 1][1]), int(convs[i + 1][4]), int(convs[i + 1][5]), int(convs[i + 1][8])
        data[w_off:w_off + kk ** 3 * cin * cout] *= np.float32(1.0 / k)
    else:
        fc = next(t for t in blob.ops if t[0] == "fc")
        data[int(fc[3]):int(fc[3]) + 3 * int(fc[2])] *= np.float32(1.0 / k)
    raw[off:] = data.tobytes()
    path = os.path.join(str(tmp_path), f"{name}_conv{i}_x{k:g}.mgw")
    with open(path, "wb") as f:
        f.write(bytes(raw))
    return path


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018"])
def test_split_fp16_transposed_convs_are_fp32_grade(capi, CG, name, tmp_path, monkeypatch):
    """Gradient calls run their 3x3x3 transposed convs on the split-fp16 kernel (conv3d_h2_kernel's gradient-pass variant:
    per-pose power-of-two scaling by the producer-recorded maximum, ConvArgs::in_amax; ReLU masks applied by the producer,
    out_mask).  Against the same call on the fp32-MFMA kernels (MI_GNINA_NO_H2_BWD at run time): same scores, atom
    gradients within 5e-6 of each pose's largest -- rigid receptor (ligand channels only), flexible rows (all channels), and
    with the gradient behind the last convolution 2^12 times larger / behind the first one 2^6 times smaller than the
    shipped model's (same scores)."""
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))

    def both(f):
        capi.set_option("MI_GNINA_NO_H2_BWD", None)
        a = f()
        with capi.option("MI_GNINA_NO_H2_BWD"):    # (a run-time switch: mi_gnina_set_option, not the environment)
            b = f()
        return a, b

    def close(a, b, key):
        B = len(a[key])
        scale = np.maximum(np.abs(b[key]).reshape(B, -1).max(1), 1e-30)
        return (np.abs(a[key] - b[key]).reshape(B, -1).max(1) / scale).max()

    s0 = capi.Scorer([name])
    s0.set_receptor(rec_xyz, rec_smt)
    shipped = s0.score_grad(poses, lig_smt)
    for model in (name, _rescaled_model(name, -1, 2.0 ** -12, tmp_path), _rescaled_model(name, 0, 2.0 ** 6, tmp_path)):
        s = capi.Scorer([capi.Model(model)])
        s.set_receptor(rec_xyz, rec_smt)
        s.enable_profile(True)
        a, b = both(lambda: s.score_grad(poses, lig_smt))
        prof = s.profile()
        s.enable_profile(False)
        rows = prof if isinstance(prof, list) else prof.get("kernels", prof)
        names = [r["kernel"] for r in rows]
        assert sum(n.startswith("convT3") and n.endswith("_h2") for n in names) == 3, names   # the split-fp16 launches ...
        assert sum(n.startswith("convT3") and not n.endswith("_h2") for n in names) == 3, names  # ... and the fp32 ones
        assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["loss"], b["loss"])
        assert np.abs(b["lig_grad"]).max() > 0 and np.abs(a["lig_grad"]).max() > 0, (model, np.abs(a["lig_grad"]).max(), np.abs(b["lig_grad"]).max(), a["loss"])
        assert close(a, b, "lig_grad") < 5e-6, (model, close(a, b, "lig_grad"))
        assert np.abs(a["pose"] - shipped["pose"]).max() < 1e-6 and close(a, shipped, "lig_grad") < 1e-5, model
        # per-pose scaling: a pose's gradient has the same bits alone (latency tiles) and in the batch
        one = s.score_grad(poses[1:2], lig_smt)
        assert np.array_equal(one["lig_grad"][0], a["lig_grad"][1]), model
        assert s.h2_fallbacks() == 0
        # flexible rows: the transposed first conv computes every channel
        rows_f = np.argsort(np.linalg.norm(rec_xyz - poses[0].mean(0), axis=1))[:12].astype(np.int32)
        flex = np.repeat(rec_xyz[rows_f][None], len(poses), 0) + np.float32(0.1)
        s.set_flex(rows_f)
        a, b = both(lambda: s.score_flex(poses, lig_smt, flex))
        assert np.array_equal(a["pose"], b["pose"])
        assert close(a, b, "lig_grad") < 5e-6 and close(a, b, "flex_grad") < 5e-6
