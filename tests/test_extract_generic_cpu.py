"""The generic TorchScript path of gnina_amd/tools/extract_weights.py (`--cnn_model file.pt` of an architecture that is not
one of the shipped families, torch_model.cpp:49-118): the program it writes, run by the CPU oracle on random grids, must
reproduce the module's own outputs -- for two custom architectures, and (where the reference's .pt files are present) for
the shipped families with the hand-written conversion switched off."""
import os

import numpy as np
import pytest
import torch

from gnina_amd.tools import extract_weights
from oracle import cnn_ref
from tests import custom_models

REF_MODELS = "/root/reference/gninasrc/lib/models"


def _check(pt_path, module, generic, C0=None, tol=2e-5):
    data, name = extract_weights.convert(pt_path, generic=generic)
    blob = cnn_ref.Blob(data)
    C0 = blob.n_rec_ch + blob.n_lig_ch
    N = int(round(blob.dimension / blob.resolution)) + 1
    rng = np.random.RandomState(3)
    grid = (rng.rand(2, C0, N, N, N) * (rng.rand(2, C0, N, N, N) < 0.1)).astype(np.float32)
    with torch.no_grad():
        want_pose, want_aff = module(torch.from_numpy(grid))
        got_pose, got_aff = cnn_ref.module_output(blob, grid)
    assert np.abs(got_pose.numpy() - want_pose.numpy()).max() < tol * max(1.0, float(want_pose.abs().max())), name
    assert np.abs(got_aff.numpy() - want_aff.numpy()).max() < tol * max(1.0, float(want_aff.abs().max())), name
    return blob


@pytest.mark.parametrize("kind", ["stack", "minidense", "postact"])
def test_custom_architectures(kind, tmp_path):
    path = str(tmp_path / f"{kind}.pt")
    m = custom_models.save_scripted(kind, path)
    blob = _check(path, m, generic=False)  # (not a shipped family: the graph walk is taken by itself)
    assert blob.meta.get("generic") == "1"
    kinds = [t[0] for t in blob.ops]
    assert kinds.count("conv") == (3 if kind == "postact" else 4) and kinds[-1] == "fc"
    if kind == "postact":  # both BatchNorms stay input transforms of the NEXT conv; nothing was folded across the ReLU
        assert [int(t[10]) >= 0 for t in blob.ops if t[0] == "conv"] == [False, True, True]
    if kind == "minidense":
        assert "gmax" in kinds and sum(1 for t in blob.ops if t[0] == "conv" and int(t[10]) >= 0) == 2  # BatchNorm kept in front of 2 convs


def test_unsupported_operator_is_named(tmp_path):
    class Bad(torch.nn.Module):
        def forward(self, x):
            y = torch.tanh(x).mean(dim=(2, 3, 4))
            return y[:, :2], y[:, 2]

    path = str(tmp_path / "bad.pt")
    torch.jit.save(torch.jit.script(Bad()), path)
    with pytest.raises(ValueError, match="unsupported operator aten::tanh"):
        extract_weights.convert(path)


def test_batchnorm_behind_relu_feeding_a_head_is_named(tmp_path):
    """conv -> ReLU -> BN -> flatten -> linear: the BatchNorm has no convolution to become the input transform of"""
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv3d(28, 4, 3, padding=1)
            self.bn = torch.nn.BatchNorm3d(4)
            self.pose = torch.nn.Linear(55296, 2)
            self.aff = torch.nn.Linear(55296, 1)

        def forward(self, x):
            x = torch.nn.functional.max_pool3d(x, 2)
            x = self.bn(torch.relu(self.c(x))).view(-1, 55296)
            return torch.log_softmax(self.pose(x), dim=1), self.aff(x).squeeze(-1)

    import json
    path = str(tmp_path / "bnhead.pt")
    torch.jit.save(torch.jit.script(M().eval()), path, _extra_files={"metadata": json.dumps({"resolution": 0.5, "dimension": 23.5})})
    with pytest.raises(ValueError, match="batch_norm"):
        extract_weights.convert(path)


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference models not present")
@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "dense"])
def test_shipped_families_through_the_graph_walk(name):
    path = os.path.join(REF_MODELS, name + ".pt")
    m = torch.jit.load(path, map_location="cpu").eval()
    blob = _check(path, m, generic=True)
    hand, _ = extract_weights.convert(path)
    ops_hand = [t[:8] for t in cnn_ref.Blob(hand).ops]
    assert [t[:8] for t in blob.ops] == ops_hand  # the same program, line for line (buffer ids, channel offsets, flags)
