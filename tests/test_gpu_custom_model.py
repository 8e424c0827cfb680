"""`--cnn_model file.pt` of an architecture that is not one of gnina's shipped families (TorchModel's constructor takes any
TorchScript module, gninasrc/lib/torch_model.cpp:49-118): the generic path of gnina_amd/tools/extract_weights.py writes the
module's program (tests/test_extract_generic_cpu.py checks it against torch on the CPU), and the engine must score it like
torch does -- forward, gradient, and on the fp32-MFMA kernels.  Architectures: tests/custom_models.py (channel counts the
shipped models do not have, a BatchNorm folded behind a convolution, a small DenseNet block with a global max pool);
goldens: tests/golden/make_custom_goldens.py (torch's own outputs on the oracle's grids)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.mark.parametrize("kind", ["stack", "minidense", "postact"])
def test_custom_architecture_scores_like_torch(capi, kind):
    G = np.load(os.path.join(GOLD, "cnn_goldens.npz"))
    W = np.load(os.path.join(GOLD, "custom_goldens.npz"))
    rec_xyz, rec_smt, lig_smt, poses = (G[f"dense_1_3/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    m = capi.Model(os.path.join(GOLD, f"custom_{kind}.mgw"))
    s = capi.Scorer([m])
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_batch(poses, lig_smt)
    scale = max(1.0, float(np.abs(W[kind + "/affinity"]).max()))
    assert np.abs(out["pose"] - W[kind + "/pose"]).max() < 1e-4
    assert np.abs(out["affinity"] - W[kind + "/affinity"]).max() < 1e-4 * scale
    one = s.score_batch(poses[1:2], lig_smt)                     # B = 1 (latency tiles) gives the batch's bits
    assert one["pose"][0] == out["pose"][1] and one["affinity"][0] == out["affinity"][1]
    s.set_precision("fp32_mfma")                                 # the fp32-MFMA program agrees
    ref = s.score_batch(poses, lig_smt)
    assert np.abs(ref["pose"] - W[kind + "/pose"]).max() < 1e-4
    assert np.abs(ref["affinity"] - W[kind + "/affinity"]).max() < 1e-4 * scale


@pytest.mark.parametrize("kind", ["stack", "minidense", "postact"])
def test_custom_architecture_gradient(capi, kind):
    """d loss / d ligand atoms against the oracle's autograd through the same program (the gradient program plans the
    transposed convolutions of whatever layers the model has)."""
    from oracle import cnn_ref
    from tests.test_gpu_gradient import oracle_lig_gradient
    G = np.load(os.path.join(GOLD, "cnn_goldens.npz"))
    rec_xyz, rec_smt, lig_smt, poses = (G[f"dense_1_3/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = os.path.join(GOLD, f"custom_{kind}.mgw")
    blob = cnn_ref.Blob(path)
    s = capi.Scorer([capi.Model(path)])
    s.set_receptor(rec_xyz, rec_smt)
    out = s.score_grad(poses[:2], lig_smt)
    fwd = s.score_batch(poses[:2], lig_smt)
    assert np.abs(out["pose"] - fwd["pose"]).max() < 5e-6
    for b in range(2):
        loss0, g0 = oracle_lig_gradient(blob, rec_xyz, rec_smt, poses[b], lig_smt)
        scale = max(np.abs(g0).max(), 1e-6)
        assert abs(out["loss"][b] - loss0) < 1e-3 * max(1.0, abs(loss0))
        assert np.abs(out["lig_grad"][b] - g0).max() < 2e-3 * scale, (kind, b, np.abs(out["lig_grad"][b] - g0).max(), scale)
