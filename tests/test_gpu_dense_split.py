"""Dense family on split-format tensors (conv3d_h2_dense.hip, round 5): gnina's default ensemble is two Dense models and a
Default2018 (gninasrc/lib/cnn_torch_scorer.cpp:28-35); a Dense model's block layers (BatchNorm -> 3x3x3 conv c_in -> 16 ->
ReLU -> concat) and 1x1x1 transitions are the reference's `module.forward` (gninasrc/lib/torch_model.cpp:185) for that family.

The forward program keeps the 24^3 / 12^3 concat buffers in the split-fp16 tensor format, folds the eval BatchNorm into the
weights and a border-class bias table, and runs conv3d_h2_d16_kernel / conv3d_h2_k1s_kernel.  Checked here:
  * layer by layer against the fp32-MFMA program of the same model (mi_debug_read_activation) -- every 16-channel slice of
    both concat buffers, so a wrong tap order / border class / octet offset names its layer;
  * scores against the reference's own TorchScript outputs (tests/golden/cnn_goldens.npz) at 1e-4;
  * two poses per workgroup on one copy of the weights (NP = 2): every pose scores the bits it scores alone;
  * the 96^3 grid of BASELINE config 5 (dense_1_3 re-gridded to 0.25 A).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def CG():
    return np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))


def _inputs(CG, name):
    return tuple(CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))


@pytest.mark.parametrize("name", ["dense", "dense_1_3"])
def test_layers_match_the_fp32_mfma_program(capi, CG, name):
    rec_xyz, rec_smt, lig_smt, poses = _inputs(CG, name)
    B = 3
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    s.enable_profile(True)
    new = s.score_batch(poses[:B], lig_smt)
    rows = s.profile()
    rows = rows if isinstance(rows, list) else rows.get("kernels", rows)
    names = [r["kernel"] for r in rows]
    s.enable_profile(False)
    # the two blocks at 24^3 / 12^3 (8 layers) and both transitions run on the split-format kernels
    assert sum(n.startswith("conv3_") and n.endswith("to16_sp_h2") for n in names) == 8, names
    assert sum(n.startswith("conv1_") and n.endswith("_sp_h2") for n in names) == 2, names
    acts = {}
    for buf in (2, 4, 6):
        acts[buf] = s.read_activation(buf, B)
    assert acts[2][1] and acts[4][1] and not acts[6][1]      # split, split, fp32 (the 6^3 block keeps round 4's kernels)
    s.set_precision("fp32_mfma")
    ref = s.score_batch(poses[:B], lig_smt)
    worst = 0.0
    for buf in (2, 4, 6):
        want, was_split = s.read_activation(buf, B)
        assert not was_split
        got = acts[buf][0]
        C = want.shape[-1]
        for c0 in range(0, C, 16):
            w, g = want[..., c0:c0 + 16], got[..., c0:c0 + 16]
            scale = max(float(np.abs(w).max()), 1e-6)
            err = float(np.abs(w - g).max()) / scale
            worst = max(worst, err)
            assert err < 2e-5, (name, buf, c0, err, scale, np.unravel_index(np.abs(w - g).argmax(), w.shape))
    print(f"{name}: worst layer deviation (relative to the slice's largest activation) {worst:.2e}")
    assert np.abs(new["pose"] - ref["pose"]).max() < 2e-5 and np.abs(new["affinity"] - ref["affinity"]).max() < 2e-5 * max(1.0, float(np.abs(ref["affinity"]).max()))
    assert np.abs(new["pose"] - CG[name + "/pose"][:B]).max() < 1e-4
    assert np.abs(new["affinity"] - CG[name + "/affinity"][:B]).max() < 1e-4 * max(1.0, float(np.abs(CG[name + "/affinity"]).max()))
    assert s.h2_fallbacks() == 0


def test_two_poses_per_workgroup_and_odd_batches(capi, CG):
    from gnina_amd import synth
    name = "dense"
    rec_xyz, rec_smt, lig_smt, poses = _inputs(CG, name)
    many = np.concatenate([poses, synth.make_poses(np.random.RandomState(5), poses[0] - poses[0].mean(0), 29)])   # 33 poses
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    big = s.score_batch(many, lig_smt)
    for b in (0, 1, 31, 32):
        one = s.score_batch(many[b:b + 1], lig_smt)
        assert one["pose"][0] == big["pose"][b] and one["affinity"][0] == big["affinity"][b], b
    with capi.option("MI_GNINA_D16_NP", 1):
        plain = s.score_batch(many, lig_smt)
    assert np.array_equal(plain["pose"], big["pose"]) and np.array_equal(plain["affinity"], big["affinity"])


def test_the_96_cubed_grid(capi, CG):
    """dense_1_3 at 0.25 A (BASELINE config 5): blocks at 48^3 / 24^3 on the split-format kernels; goldens = the reference's
    own dense_1.3.pt on the oracle's 96^3 grids (tests/golden/cnn_goldens_96.npz)."""
    G96 = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens_96.npz"))
    name = "dense_1_3"
    rec_xyz, rec_smt, lig_smt, poses = _inputs(CG, name)
    m = capi.Model(name, resolution=0.25, dimension=23.75)
    assert m.grid_points == 96
    s = capi.Scorer([m])
    s.set_receptor(rec_xyz, rec_smt)
    s.enable_profile(True)
    out = s.score_batch(poses[:2], lig_smt)
    rows = s.profile()
    rows = rows if isinstance(rows, list) else rows.get("kernels", rows)
    names = [r["kernel"] for r in rows]
    assert sum(n.endswith("to16_sp_h2") for n in names) == 8, names
    assert np.abs(out["pose"] - G96[name + "/pose"]).max() < 1e-4
    assert np.abs(out["affinity"] - G96[name + "/affinity"]).max() < 1e-4
