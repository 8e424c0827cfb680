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


def _write_atoms(path, rec_xyz, rec_smt, lig_smt, poses, names):
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", len(rec_smt), len(lig_smt), len(poses), len(names)))
        for n in names:
            f.write(struct.pack("<i", len(n)) + n.encode())
        f.write(rec_xyz.astype("<f4").tobytes())
        f.write(rec_smt.astype("<i4").tobytes())
        f.write(lig_smt.astype("<i4").tobytes())
        f.write(poses.astype("<f4").tobytes())


@pytest.mark.gpu
def test_torchmodel_seam_gradients_and_rotate(exe, golden_dir, tmp_path):
    """HipTorchModel (torch_model.h:32-46): forward(compute_gradient) + getLigandGradient / getReceptorGradient for
    EVERY receptor atom, and forward(rotate) with a seeded rotation stream."""
    from gnina_amd import capi
    G = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = tmp_path / "atoms.bin"
    _write_atoms(path, rec_xyz, rec_smt, lig_smt, poses[:1], [name])
    r = subprocess.run([exe, str(path), WEIGHTS, "--torchmodel"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.strip().split("\n")]
    fwd = [l for l in lines if l[0] == "tm_forward"][0]
    assert abs(float(fwd[1]) - G[name + "/pose"][0]) < 1e-4 and abs(float(fwd[2]) - G[name + "/affinity"][0]) < 1e-4
    grad = [l for l in lines if l[0] == "tm_grad"][0]
    assert float(grad[1]) == float(fwd[1]) and int(grad[5]) == len(lig_smt) and int(grad[7]) == len(rec_smt)
    capi.init(0)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    ref = s.score_grad(poses[:1], lig_smt)
    gl = np.array([[float(x) for x in l[2:5]] for l in lines if l[0] == "gl"])
    assert np.abs(gl - ref["lig_grad"][0]).max() <= 1e-6 * max(1.0, np.abs(gl).max())
    s.set_flex(np.arange(len(rec_smt)))
    full = s.score_flex(poses[:1], lig_smt, rec_xyz[None])
    gr = {int(l[1]): [float(x) for x in l[2:5]] for l in lines if l[0] == "gr"}
    assert len(gr) > 10
    for i, g in gr.items():
        assert np.abs(np.array(g) - full["flex_grad"][0, i]).max() <= 1e-6 * max(1.0, np.abs(full["flex_grad"]).max())
    rot = [l for l in lines if l[0] == "tm_rotated"][0]
    q = np.array([float(x) for x in rot[5:9]], dtype=np.float32)
    assert abs(np.linalg.norm(q) - 1) < 1e-5
    s2 = capi.Scorer([name])
    s2.set_receptor(rec_xyz, rec_smt)
    s2.set_rotations(q[None])
    o = s2.score_batch(poses[:1], lig_smt)
    assert abs(float(rot[1]) - o["pose"][0]) < 1e-6 and abs(float(rot[1]) - float(fwd[1])) > 1e-6


@pytest.mark.gpu
def test_cnn_rotations_average_like_the_reference(exe, golden_dir, tmp_path):
    """--cnn_rotation N (cnn_torch_scorer.cpp:117-193): mean score / affinity / loss over models x orientations,
    population variance of all affinities, mean forces; orientation 0 is the pose as it is."""
    from gnina_amd import capi
    G = np.load(os.path.join(golden_dir, "cnn_goldens.npz"))
    names = ["crossdock_default2018", "crossdock_default2018_KD_4"]
    base = names[0]
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = tmp_path / "atoms.bin"
    _write_atoms(path, rec_xyz, rec_smt, lig_smt, poses[:1], names)
    n_rot = 5
    r = subprocess.run([exe, str(path), WEIGHTS, "--rotations", str(n_rot), "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.strip().split("\n")]
    quats = np.array([[float(x) for x in l[2:6]] for l in lines if l[0] == "quat"], dtype=np.float32)
    assert quats.shape == (n_rot, 4) and list(quats[0]) == [1, 0, 0, 0]
    score, aff, loss, var = (float(x) for x in [l for l in lines if l[0] == "rot_score"][0][1:5])
    capi.init(0)
    per_model = []
    grads = []
    for n in names:
        s = capi.Scorer([n])
        s.set_receptor(rec_xyz, rec_smt)
        s.set_rotations(quats)
        o = s.score_grad(np.repeat(poses[:1], n_rot, 0), lig_smt)
        per_model.append(o)
        grads.append(o["lig_grad"])
    p = np.stack([o["pose"] for o in per_model])          # [model][rotation]
    a = np.stack([o["affinity"] for o in per_model])
    assert abs(score - p.astype(np.float64).mean()) < 2e-6 and abs(aff - a.mean()) < 2e-5
    assert abs(var - a.var()) < 1e-4
    forces = np.array([[float(x) for x in l[2:5]] for l in lines if l[0] == "force"])
    want = np.mean(grads, axis=(0, 1))
    heavy = lig_smt > 1
    assert np.abs(forces[heavy] - want[heavy]).max() <= 2e-3 * max(1e-6, np.abs(want).max())
    again = [l for l in lines if l[0] == "rot_again"][0]
    assert float(again[1]) == score and abs(float(again[3]) - var) < 1e-7
    # rotation 0 alone is the plain score
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    assert abs(s.score_batch(poses[:1], lig_smt)["pose"][0] - p[:, 0].mean()) < 1e-6


def _synthetic_pdbqt_pair(tmp_path):
    """A receptor of random typed atoms around a pocket and a flexible chain ligand, as PDBQT files."""
    from tests import ref_cases as RC
    rng = np.random.RandomState(3)
    ad = ["C", "A", "N", "NA", "OA", "S", "HD"]
    lines = []
    lig_text = RC.long_chain_ligand(n=10, origin=(0.0, 0.0, 0.0))
    k = 0
    while k < 900:
        p = rng.uniform(-16, 22, 3)
        if abs(p[1]) < 3.0 and abs(p[2]) < 3.0 and -3 < p[0] < 15:      # keep a channel for the ligand
            continue
        lines.append(RC.atom_line(k + 1, "X", *p, ad[rng.randint(len(ad))], res="REC", resnum=1 + k // 8))
        k += 1
    rec = tmp_path / "rec.pdbqt"
    lig = tmp_path / "lig.pdbqt"
    rec.write_text("\n".join(lines) + "\n")
    lig.write_text(lig_text)
    return str(rec), str(lig)


@pytest.mark.gpu
def test_reference_code_runs_on_the_hip_igrid(tmp_path):
    """oracle/_ref/test_igrid_dropin: gnina's OWN model::eval_deriv, quasi_newton (CPU bfgs<>) and monte_carlo with
    HipCache as their igrid, next to the same calls on gnina's cache; HipQuasiNewton's device dispatch."""
    exe = os.path.join(ROOT, "oracle", "_ref", "test_igrid_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_igrid_dropin is built where /root/reference exists (make -f oracle/Makefile.ref dropin)")
    rec, lig = _synthetic_pdbqt_pair(tmp_path)
    r = subprocess.run([exe, rec, lig], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.strip().split("\n")]
    ev = [l for l in lines if l[0] == "eval_deriv"]
    assert len(ev) == 4
    for l in ev:
        e1, e2, dch, scale, i1, i2 = float(l[2]), float(l[3]), float(l[5]), float(l[7]), float(l[9]), float(l[10])
        assert abs(e1 - e2) <= 1e-4 * max(1.0, abs(e1)) and dch <= 1e-3 * max(1.0, scale)
        assert abs(i1 - i2) <= 1e-4 * max(1.0, abs(i1))
    cpu = [l for l in lines if l[0] == "cpu_bfgs_on"]
    close = sum(abs(float(l[3]) - float(l[5])) <= 0.05 * abs(float(l[3])) + 0.5 for l in cpu)
    assert len(cpu) == 3 and close >= 2                      # gnina's bfgs<> reaches the same minima on either igrid
    dev = [l for l in lines if l[0] == "device_bfgs"]
    assert all(l[3] == "1" for l in dev) and all(np.isfinite(float(l[5])) for l in dev)
    assert abs(float(dev[0][5]) - float(cpu[0][3])) <= 0.05 * abs(float(cpu[0][3])) + 0.5
    assert all(l[2] == "0" for l in lines if l[0] == "device_bfgs_on_ref_cache")   # not a HipCache: caller's CPU path
    mc = {l[1]: l for l in lines if l[0] == "mc_on"}
    assert int(mc["ref_cache"][3]) >= 1 and int(mc["hip_cache"][3]) >= 1
    assert np.isfinite(float(mc["hip_cache"][5]))


@pytest.mark.gpu
def test_reference_cnn_code_runs_on_the_hip_scorer(tmp_path):
    """oracle/_ref/test_cnn_dropin: the PRIMARY seam against gnina's real headers.  HipCNNScorer derives from gnina's
    own DLScorer (dl_scorer.h) and uses gnina's own setLigand / setReceptor (dl_scorer.cpp, unmodified); gnina's
    non_cache_cnn::eval / eval_deriv (non_cache_cnn.cpp:33-169), model::eval_deriv (the CNN gradient folded by gnina's
    tree code) and quasi_newton inside refine_structure's loop run on it -- next to the batched C-ABI entry points
    (mi_cnn_eval_batch, mi_cnn_refine_batch, mi_scorer_score_batch) on the same real complex (GSK3B + adduct ligand)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "test_cnn_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/test_cnn_dropin is built where /root/reference exists (make -f oracle/Makefile.ref dropin_cnn)")
    F = np.load(os.path.join(ROOT, "tests", "golden", "real_complex.npz"))
    rec, lig = tmp_path / "rec.pdbqt", tmp_path / "lig.pdbqt"
    rec.write_bytes(bytes(F["rec_pdbqt"]))
    lig.write_bytes(bytes(F["lig_adduct_pdbqt"]))
    r = subprocess.run([exe, str(rec), str(lig), os.path.join(ROOT, "gnina_amd", "weights"), "default2017"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.strip().split("\n")]
    ev = [l for l in lines if l[0] == "cnn_eval"]
    assert len(ev) == 4
    for l in ev:
        g_eval, a_eval, g_der, a_der, dch, scale = (float(l[i]) for i in (3, 5, 7, 9, 11, 13))
        assert abs(g_eval - a_eval) <= 1e-4 * max(1.0, abs(g_eval)), l        # non_cache_cnn::eval
        assert abs(g_der - a_der) <= 1e-4 * max(1.0, abs(g_der)), l          # non_cache_cnn::eval_deriv (energy)
        assert dch <= 2e-3 * max(1e-3, scale), l                              # ... folded into change by gnina's tree code
    assert float(ev[3][3]) > float(ev[0][3]) + 1.0                            # the out-of-box pose pays the slope penalty
    sc = [l for l in lines if l[0] == "score"][0]
    seam, abi = [float(x) for x in sc[2:5]], [float(x) for x in sc[6:9]]
    assert max(abs(a - b) for a, b in zip(seam, abi)) <= 1e-5                 # DLScorer::score == the batched entry point
    rf = [l for l in lines if l[0] == "refine"][0]
    start, on_seam, on_abi = float(rf[2]), float(rf[4]), float(rf[8])
    assert np.isfinite(on_seam) and np.isfinite(on_abi)
    assert on_seam < start and on_abi < start                                 # both minimise the CNN loss from the same start
    assert abs(on_seam - on_abi) <= 0.1 * abs(start) + 0.05                   # ... to equivalent minima (CNN gradients amplify)
    fc = [l for l in lines if l[0] == "fresh_copy"][0]
    assert float(fc[1]) == float(fc[4]) and float(fc[2]) == float(fc[5])      # fresh_copy shares the device weights


@pytest.mark.gpu
def test_multi_device_pool_from_cpp():
    """mi_pool (the C++ multi-device host path): tests/cpp/test_pool.cpp creates a pool over every visible
    GPU from C++ (worker thread + scorer per device, contiguous pose shards) and scores one batch through the host-buffer
    path and the device-resident path (RCCL scatter / gather over xGMI when there is more than one device): both must
    equal the single-scorer result bit for bit.  One GPU here: the pool forwards to its single scorer; with N > 1
    visible the same binary exercises the sharded paths and prints strong-scaling rates."""
    exe = os.path.join(ROOT, "gnina_amd", "lib", "test_pool")
    assert os.path.exists(exe), "run `python __graft_entry__.py build`"
    r = subprocess.run([exe, os.path.join(ROOT, "gnina_amd", "weights"), "0", "600"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [l.split() for l in r.stdout.strip().split("\n") if l.startswith("pool ")]
    assert rows, r.stdout
    for l in rows:
        assert l[6] == "1" and l[10] == "1", l          # host_path_equal, device_path_equal
    n_vis = int([l for l in r.stdout.split("\n") if l.startswith("devices_visible")][0].split()[1])
    assert int(rows[-1][2]) == n_vis                    # the largest pool spans every visible device


@pytest.mark.gpu
def test_pool_python_binding_matches_single_scorer():
    from gnina_amd import capi, synth
    capi.init(0)
    m = capi.Model("crossdock_default2018")
    rng = np.random.RandomState(3)
    rx, rs = synth.make_receptor(rng, 1500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 24, synth.mapped_types(m.chan_of_smt(True)))
    poses = synth.make_poses(rng, lx, 77)
    s = capi.Scorer([m])
    s.set_receptor(rx, rs)
    want = s.score_batch(poses, ls)
    p = capi.Pool(["crossdock_default2018"])
    p.set_receptor(rx, rs)
    got = p.score_batch(poses, ls)
    for k in ("pose", "affinity", "loss"):
        assert np.array_equal(want[k], got[k]), k
    # ragged (virtual-screen) seam through the pool
    Lmax = 30
    xyz = np.zeros((40, Lmax, 3), np.float32)
    smt = np.full((40, Lmax), -1, np.int32)
    for b in range(40):
        L = rng.randint(10, Lmax + 1)
        a, t = synth.make_ligand(rng, L, synth.mapped_types(m.chan_of_smt(True)))
        xyz[b, :L], smt[b, :L] = a, t
    w2, g2 = s.score_ragged(xyz, smt), p.score_ragged(xyz, smt)
    assert np.array_equal(w2["pose"], g2["pose"]) and np.array_equal(w2["affinity"], g2["affinity"])
    assert p.info()["ranks"] == len(p.devices)


@pytest.mark.gpu
def test_pool_shards_with_several_workers_on_one_gpu(allow_duplicate_devices):
    """Every shard boundary and offset of mi_pool's host-buffer and device-resident paths, on a one-GPU box: the test hook
    MI_POOL_ALLOW_DUPLICATE_DEVICES lets devices = [0, 0, 0] run three workers (three scorers, three streams) on the same
    GPU; the device-resident path then moves the shards with device-to-device copies (RCCL refuses duplicate devices --
    with distinct devices the same shards travel through ncclSend / ncclRecv).  Results must equal one scorer's bits."""
    from gnina_amd import capi, synth
    capi.init(0)
    name = "default2017"
    m = capi.Model(name)
    rng = np.random.RandomState(8)
    rx, rs = synth.make_receptor(rng, 1800, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 20, synth.mapped_types(m.chan_of_smt(True)))
    B = 203                                                       # 67 / 68 / 68: uneven contiguous shards
    poses = synth.make_poses(rng, lx, B)
    centers = (poses.mean(1) + rng.uniform(-0.5, 0.5, (B, 3))).astype(np.float32)
    s = capi.Scorer([m])
    s.set_receptor(rx, rs)
    want = s.score_batch(poses, ls, centers=centers)
    p = capi.Pool([name], devices=[0, 0, 0])
    p.set_receptor(rx, rs)
    got = p.score_batch(poses, ls, centers=centers)
    for k in ("pose", "affinity", "loss"):
        assert np.array_equal(want[k], got[k]), k
    # device buffers straight from the HIP runtime (no torch in this test: plain C ABI + hipMalloc, like a C++ caller)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes, hip.hipMemcpy.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def dmalloc(nbytes):
        ptr = C.c_void_p()
        assert hip.hipSetDevice(0) == 0 and hip.hipMalloc(C.byref(ptr), nbytes) == 0
        return ptr

    d_lig, d_cen, d_out = dmalloc(poses.nbytes), dmalloc(centers.nbytes), dmalloc(4 * B * 4)
    assert hip.hipMemcpy(d_lig, poses.ctypes.data_as(C.c_void_p), poses.nbytes, 1) == 0       # hipMemcpyHostToDevice
    assert hip.hipMemcpy(d_cen, centers.ctypes.data_as(C.c_void_p), centers.nbytes, 1) == 0
    outp = [C.c_void_p(d_out.value + 4 * B * a) for a in range(4)]
    p.score_batch_device(d_lig, ls, B, poses.shape[1], outp[0], outp[1], outp[2], outp[3], centers_ptr=d_cen)
    o = np.empty((4, B), dtype=np.float32)
    assert hip.hipMemcpy(o.ctypes.data_as(C.c_void_p), d_out, o.nbytes, 2) == 0                # hipMemcpyDeviceToHost
    for ptr in (d_lig, d_cen, d_out):
        hip.hipFree(ptr)
    assert np.array_equal(o[0], want["pose"]) and np.array_equal(o[1], want["affinity"]) and np.array_equal(o[2], want["loss"])
    info = p.info()
    assert info["ranks"] == 3 and info["device_path_transport"] == "copies" and info["calls_device_path"] == 1
    with pytest.raises(capi.MiGninaError):                        # without the hook a duplicate is an error
        capi.set_option("MI_POOL_ALLOW_DUPLICATE_DEVICES", None)
        capi.Pool([name], devices=[0, 0])
