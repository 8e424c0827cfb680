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


@pytest.mark.gpu
def test_dlscorer_adapter_flexible_residues(exe, golden_dir, tmp_path):
    """Flexible-residue atoms in front of the ligand (dl_scorer.cpp:93-193): refreshed coordinates are scored
    and score(compute_gradient=true) leaves their forces in minus_forces (cnn_torch_scorer.cpp:216-224)."""
    from gnina_amd import capi
    G = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    name = "crossdock_default2018"
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    # put the receptor atoms closest to the ligand first: they become the flexible rows; no hydrogens, so
    # add_minus_forces' heavy-atom counter (model.cu:247-259) walks the gradient one to one
    order = np.argsort(np.linalg.norm(rec_xyz - poses[0].mean(axis=0), axis=1), kind="stable")
    rec_xyz, rec_smt = rec_xyz[order], rec_smt[order]
    K = 10
    assert (rec_smt[:K] > 1).all() and (lig_smt > 1).all()
    path = tmp_path / "atoms.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), 1, 1))
        f.write(struct.pack("<i", len(name)) + name.encode())
        f.write(rec_xyz.astype("<f4").tobytes())
        f.write(rec_smt.astype("<i4").tobytes())
        f.write(lig_smt.astype("<i4").tobytes())
        f.write(poses[:1].astype("<f4").tobytes())
    r = subprocess.run([exe, str(path), WEIGHTS, "--flex", str(K)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.strip().split("\n")]
    rest = [l for l in lines if l[0] == "flex_rest"][0]
    moved = [l for l in lines if l[0] == "flex_moved"][0]
    forces = np.array([[float(x) for x in l[2:5]] for l in lines if l[0] == "force"])
    capi.init(0)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    assert abs(float(rest[1]) - s.score_batch(poses[:1], lig_smt)["pose"][0]) < 1e-6
    s.set_flex(np.arange(K))
    flex = rec_xyz[:K].copy()
    flex[:, 0] += 0.25
    out = s.score_flex(poses[:1], lig_smt, flex[None])
    assert float(moved[1]) == out["pose"][0] and float(moved[1]) != float(rest[1])
    assert np.abs(forces[:K] - out["flex_grad"][0]).max() < 1e-7
    assert np.abs(forces[K:] - out["lig_grad"][0]).max() < 1e-7
    assert np.abs(forces[:K]).max() > 0


@pytest.mark.gpu
def test_dlscorer_adapter_covalent_branch(exe, golden_dir, tmp_path):
    """Without a ligand the `iscov` atoms are the CNN's ligand (DLScorer::setLigand, dl_scorer.cpp:43-69)."""
    G = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = tmp_path / "atoms.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), 1, 1))
        f.write(struct.pack("<i", len(name)) + name.encode())
        f.write(rec_xyz.astype("<f4").tobytes())
        f.write(rec_smt.astype("<i4").tobytes())
        f.write(lig_smt.astype("<i4").tobytes())
        f.write(poses[:1].astype("<f4").tobytes())
    r = subprocess.run([exe, str(path), WEIGHTS, "--cov"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    cov = [l.split() for l in r.stdout.strip().split("\n") if l.startswith("cov")][0]
    assert abs(float(cov[1]) - G[name + "/pose"][0]) < 1e-4 and abs(float(cov[2]) - G[name + "/affinity"][0]) < 1e-4
