"""The C++ host adapters (gnina_amd/host: HipCNNScorer : DLScorer, HipTorchModel) driven the way
gnina drives CNNTorchScorer (tests/cpp/test_host_scorer.cpp)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")


@pytest.fixture(scope="module")
def exe():
    from gnina_amd import build
    return build.build_host()


def test_builtin_model_names(exe):
    out = subprocess.run([exe, "x", WEIGHTS, "--names"], capture_output=True, text=True, check=True).stdout.split()
    assert "default2017" in out and "dense_1_3" in out and "crossdock_default2018_KD_4" in out


@pytest.mark.gpu
def test_dlscorer_adapter_matches_goldens(exe, golden_dir, tmp_path):
    G = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]  # = gnina's default ensemble
    base = names[0]
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = tmp_path / "atoms.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), len(poses), 0))  # 0 names -> default ensemble
        f.write(rec_xyz.astype("<f4").tobytes())
        f.write(rec_smt.astype("<i4").tobytes())
        f.write(lig_smt.astype("<i4").tobytes())
        f.write(poses.astype("<f4").tobytes())
    r = subprocess.run([exe, str(path), WEIGHTS], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0].startswith("usage_error_ok Invalid model name")
    assert lines[1] == "models 3 initialized 1 has_affinity 1"
    pose_ref = np.mean([G[n + "/pose"] for n in names], axis=0)
    affs = np.stack([G[n + "/affinity"] for n in names])
    single = [l.split() for l in lines if l.startswith("single")]
    batch = [l.split() for l in lines if l.startswith("batch")]
    assert len(single) == len(batch) == len(poses)
    for b in range(len(poses)):
        s, a, l, v, s_copy = (float(single[b][i]) for i in (2, 3, 4, 5, 7))
        assert abs(s - pose_ref[b]) < 1e-4 and abs(a - affs[:, b].mean()) < 1e-4
        assert abs(v - affs[:, b].var()) < 1e-4
        assert s_copy == s                                   # fresh_copy() scores identically
        assert float(batch[b][2]) == s and float(batch[b][3]) == a   # batched == one at a time, bitwise
    box = [l for l in lines if l.startswith("box")][0].split()
    assert abs((float(box[2]) - float(box[1])) - 23.5) < 1e-4 and box[3] == "47"
