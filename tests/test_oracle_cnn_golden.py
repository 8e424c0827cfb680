"""Pin the CNN oracle (oracle/cnn_ref.py + the MIGNINA1 weight blobs) against outputs of the
reference's own TorchScript models (tests/golden/cnn_goldens.npz, made by make_cnn_goldens.py
from /root/reference/gninasrc/lib/models/*.pt), and -- when the reference tree is present --
directly against torch.jit.load of those files.  Tolerance: 1e-4 abs on pose and affinity
(BASELINE.json north_star), i.e. tighter than the reference's own CPU-vs-GPU check (3 decimals,
test/gnina/test_cnn.py:43)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import cnn_ref, voxel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")
MODELS = ["default2017", "crossdock_default2018", "dense", "dense_1_3", "dense_1_3_PT_KD_3",
          "crossdock_default2018_KD_4"]


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "cnn_goldens.npz"))


def oracle_grids(blob, G, name):
    rmap = voxel.typer_parse(blob.recmap_text())
    lmap = voxel.typer_parse(blob.ligmap_text())
    poses = G[name + "/poses"]
    return np.stack([voxel.voxelize_pose(G[name + "/rec_xyz"], G[name + "/rec_smt"], poses[b], G[name + "/lig_smt"],
                                         rmap, lmap, None, blob.resolution, blob.dimension,
                                         blob.radius_scaling)[0] for b in range(len(poses))])


@pytest.mark.parametrize("name", MODELS)
def test_blob_metadata(name):
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    assert blob.resolution == 0.5 and blob.dimension == 23.5
    assert voxel.grid_points(blob.resolution, blob.dimension) == 48
    if name == "default2017":
        assert (blob.n_rec_ch, blob.n_lig_ch) == (16, 19)
    else:
        assert (blob.n_rec_ch, blob.n_lig_ch) == (14, 14)
    # merged channels of the 2018 maps (torch_model.cpp:16-46)
    if name != "default2017":
        rmap, _ = voxel.typer_parse(blob.recmap_text())
        names = voxel.smina_type_names()
        assert len({rmap[names.index(n)] for n in ("Bromine", "Iodine", "Chlorine", "Fluorine")}) == 1


@pytest.mark.parametrize("name", MODELS)
def test_oracle_scores_match_reference_outputs(G, name):
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    grids = oracle_grids(blob, G, name)
    np.testing.assert_allclose(grids.reshape(len(grids), -1).sum(1, dtype=np.float64), G[name + "/grid_sum"],
                               rtol=1e-6)
    with torch.no_grad():
        pose, aff, loss = cnn_ref.scores(blob, grids)
    assert np.abs(pose.numpy() - G[name + "/pose"]).max() < 1e-4
    assert np.abs(aff.numpy() - G[name + "/affinity"]).max() < 1e-4
    assert np.abs(loss.numpy() - G[name + "/loss"]).max() < 1e-3 * max(1.0, G[name + "/loss"].max())
    # fp32 headroom: fp64 evaluation of the reference agrees to ~1e-5
    assert np.abs(pose.numpy() - G[name + "/pose64"]).max() < 1e-4
    assert np.abs(aff.numpy() - G[name + "/affinity64"]).max() < 1e-4


@pytest.mark.skipif(not os.path.isdir("/root/reference/gninasrc/lib/models"), reason="reference tree absent")
def test_oracle_vs_reference_torchscript_live():
    """Direct check against the reference's TorchScript (only where /root/reference exists)."""
    torch.manual_seed(0)
    for name, stem in (("default2017", "default2017"), ("crossdock_default2018", "crossdock_default2018"),
                       ("dense", "dense")):
        blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
        m = torch.jit.load(f"/root/reference/gninasrc/lib/models/{stem}.pt", map_location="cpu")
        C = blob.n_rec_ch + blob.n_lig_ch
        x = torch.rand(1, C, 48, 48, 48) * (torch.rand(1, C, 48, 48, 48) < 0.08)
        with torch.no_grad():
            logp, aff = m(x)
            lg, a2 = cnn_ref.forward_logits(blob, x)
        assert (torch.log_softmax(lg, 1) - logp).abs().max() < 2e-4 * max(1.0, logp.abs().max().item())
        assert (aff - a2).abs().max() < 1e-4 * max(1.0, aff.abs().max().item())


def test_overlap_toy_models_match_reference_goldens(golden_dir):
    """test/gnina/data/overlap*.pt (skip_softmax + apply_logistic_loss): oracle vs the reference modules' outputs."""
    import os
    import numpy as np
    from oracle import cnn_ref, voxel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    OG = np.load(os.path.join(golden_dir, "overlap_goldens.npz"))
    for name in ("overlap", "overlap_smallr"):
        blob = cnn_ref.Blob(os.path.join(root, "gnina_amd", "weights", name + ".mgw"))
        assert blob.skip_softmax and blob.apply_logistic_loss and blob.family == "Overlap"
        rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
        assert rmap[1] == 1 and lmap[1] == 1            # every heavy type on one line -> one channel each
        for case in ("C_C1", "CC_CC2", "rand8", "far"):
            k = f"{name}/{case}/"
            rec, lig = OG[k + "rec"], OG[k + "lig"]
            t = np.full(len(rec), 2, np.int32), np.full(len(lig), 2, np.int32)
            grid, _ = voxel.voxelize_pose(rec, t[0], lig, t[1], rmap, lmap, None, blob.resolution, blob.dimension,
                                          blob.radius_scaling)
            p, a, l = cnn_ref.scores(blob, grid[None])
            assert abs(float(p[0]) - OG[k + "pose"]) <= 1e-5 * OG[k + "pose"] + 1e-30
            assert abs(float(l[0]) - OG[k + "loss"]) < 1e-4 and float(a[0]) == 0.0
