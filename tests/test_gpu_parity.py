"""GPU parity tests (MI355X): the HIP path, called through the C ABI, against
  (a) the reference's own goldens (gninagrid grids; TorchScript outputs in tests/golden/*.npz),
  (b) the CPU oracle (oracle/) on the same seeded inputs.
Tolerances: voxel grids 1e-5 abs vs the oracle (reference's own: 1e-4, compare_dx.py:24), support
set and grid centres bit-exact; CNN pose / affinity 1e-4 abs (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import cnn_ref, voxel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")
MODELS = ["default2017", "crossdock_default2018", "dense", "dense_1_3", "dense_1_3_PT_KD_3",
          "crossdock_default2018_KD_4"]
ALIPH_C = 2


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def VG(golden_dir):
    return np.load(os.path.join(golden_dir, "voxel_goldens.npz"))


@pytest.fixture(scope="module")
def CG(golden_dir):
    return np.load(os.path.join(golden_dir, "cnn_goldens.npz"))


def oracle_maps(blob):
    return voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())


def test_library_loaded_and_device(capi):
    assert capi.lib().mi_gnina_device_count() >= 1
    assert capi.lib().mi_gnina_abi_version() == capi.ABI_VERSION


def test_typer_bit_exact(capi):
    for name in ("default2017", "crossdock_default2018"):
        m = capi.Model(name)
        blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
        rmap, lmap = oracle_maps(blob)
        assert np.array_equal(m.chan_of_smt(False), rmap[0]) and m.n_rec_channels == rmap[1]
        assert np.array_equal(m.chan_of_smt(True), lmap[0]) and m.n_lig_channels == lmap[1]
        radii = np.array([m.type_channel(False, t)[1] for t in range(28)], dtype=np.float32)
        assert np.array_equal(radii, voxel.xs_radii())


def test_voxelize_reference_golden_cc(capi, VG):
    """gninagrid golden (test/gninagrid/CMakeLists.txt:26-28): CC.xyz as receptor and ligand, default2017 maps."""
    s = capi.Scorer(["default2017"])
    lig = VG["cc_xyz"]
    smt = np.full(len(lig), ALIPH_C, dtype=np.int32)
    rec = np.round(lig.astype(np.float64), 3).astype(np.float32)
    s.set_receptor(rec, smt)
    grids, cen = s.voxelize_batch(lig[None], smt)
    assert grids.shape == (1, 35, 48, 48, 48)
    np.testing.assert_allclose(cen[0] - 11.75, VG["ccdx_lig_origin"], atol=1e-5)
    assert np.abs(grids[0, 16] - VG["ccdx_lig"]).max() < 1e-4
    assert np.abs(grids[0, 0] - VG["ccdx_rec"]).max() < 1e-4
    assert np.abs(grids[0, 16] - VG["ccmap_lig"]).max() < 1e-4
    assert not np.delete(grids[0], [0, 16], axis=0).any()


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018"])
def test_voxelize_vs_oracle_synthetic(capi, CG, name):
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = oracle_maps(blob)
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    grids, cen = s.voxelize_batch(poses, lig_smt)
    for b in range(len(poses)):
        ref, c = voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap)
        assert np.array_equal(c, cen[b])                       # grid centre bit-exact
        assert np.array_equal(ref != 0, grids[b] != 0)          # grid indexing / support set bit-exact
        assert np.abs(ref - grids[b]).max() < 1e-5
    np.testing.assert_allclose(grids.reshape(len(poses), -1).sum(1, dtype=np.float64), CG[name + "/grid_sum"],
                               rtol=1e-6)


def test_voxelize_edge_cases(capi):
    name = "crossdock_default2018"
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = oracle_maps(blob)
    rng = np.random.RandomState(7)
    rec_xyz = rng.uniform(-12, 12, (300, 3)).astype(np.float32)
    rec_smt = rng.randint(0, 28, 300).astype(np.int32)          # includes hydrogens / unmapped types
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    # (a) ligand with hydrogens (untyped rows count for the centre), explicit centres mixed with NaN
    lig_smt = np.array([0, 1, 2, 6, 10, 17, 0, 14], dtype=np.int32)
    poses = rng.normal(0, 3, (3, len(lig_smt), 3)).astype(np.float32)
    centers = np.array([[np.nan] * 3, [1.0, -2.0, 0.5], [np.nan] * 3], dtype=np.float32)
    grids, cen = s.voxelize_batch(poses, lig_smt, centers)
    for b in range(3):
        cin = None if np.isnan(centers[b, 0]) else centers[b]
        ref, c = voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, cin)
        assert np.array_equal(c, cen[b])
        assert np.array_equal(ref != 0, grids[b] != 0)
        assert np.abs(ref - grids[b]).max() < 1e-5
    # (b) atoms straddling the box boundary and far outside
    far = poses[:1].copy()
    far[0, :, 0] += 11.0
    g2, c2 = s.voxelize_batch(far, lig_smt, np.zeros((1, 3), dtype=np.float32))
    ref, _ = voxel.voxelize_pose(rec_xyz, rec_smt, far[0], lig_smt, rmap, lmap, np.zeros(3, dtype=np.float32))
    assert np.array_equal(ref != 0, g2[0] != 0) and np.abs(ref - g2[0]).max() < 1e-5
    # (c) empty batch and a ligand with no typed atom at all
    g0, _ = s.voxelize_batch(np.zeros((0, 4, 3), dtype=np.float32), np.zeros(4, dtype=np.int32))
    assert g0.shape[0] == 0
    hs = np.zeros(3, dtype=np.int32)
    g3, c3 = s.voxelize_batch(poses[:1, :3], hs)
    ref, c = voxel.voxelize_pose(rec_xyz, rec_smt, poses[0, :3], hs, rmap, lmap)
    assert np.array_equal(c, c3[0]) and np.abs(ref - g3[0]).max() < 1e-5
    assert not g3[0, rmap[1]:].any()


@pytest.mark.parametrize("name", MODELS)
def test_scores_match_reference_torchscript_goldens(capi, CG, name):
    """End to end (voxelize + CNN) vs outputs of the reference's own .pt on the same atoms."""
    s = capi.Scorer([name])
    s.set_receptor(CG[name + "/rec_xyz"], CG[name + "/rec_smt"])
    out = s.score_batch(CG[name + "/poses"], CG[name + "/lig_smt"])
    assert np.abs(out["pose"] - CG[name + "/pose"]).max() < 1e-4
    assert np.abs(out["affinity"] - CG[name + "/affinity"]).max() < 1e-4
    assert np.abs(out["loss"] - CG[name + "/loss"]).max() < 1e-3
    assert not out["variance"].any()                    # single model -> variance 0


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "dense"])
def test_cnn_forward_on_given_grids_vs_oracle(capi, name):
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    C = blob.n_rec_ch + blob.n_lig_ch
    rng = np.random.RandomState(3)
    grids = (rng.rand(3, C, 48, 48, 48) * (rng.rand(3, C, 48, 48, 48) < 0.08)).astype(np.float32)
    s = capi.Scorer([name])
    pose, aff, loss = s.forward_grids(grids)
    with torch.no_grad():
        p0, a0, l0 = cnn_ref.scores(blob, grids)
        p64, a64, _ = cnn_ref.scores(blob, grids, torch.float64)
    # random dense-valued grids give large logits: compare against fp64 with a relative budget
    scale = max(1.0, float(a64.abs().max()))
    assert np.abs(pose - p64.numpy()).max() < 1e-4
    assert np.abs(aff - a64.numpy()).max() < 1e-4 * scale
    assert np.abs(aff - a0.numpy()).max() < 2e-4 * scale


def test_ensemble_mean_and_variance(capi, CG):
    names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]  # gnina's default ensemble
    s = capi.Scorer(names)
    base = names[0]
    s.set_receptor(CG[base + "/rec_xyz"], CG[base + "/rec_smt"])
    out = s.score_batch(CG[base + "/poses"], CG[base + "/lig_smt"])
    pose = np.mean([CG[n + "/pose"] for n in names], axis=0)
    affs = np.stack([CG[n + "/affinity"] for n in names])
    assert np.abs(out["pose"] - pose).max() < 1e-4
    assert np.abs(out["affinity"] - affs.mean(0)).max() < 1e-4
    assert np.abs(out["variance"] - affs.var(0)).max() < 1e-4      # population variance (cnn_torch_scorer.cpp:181-191)
    for i, n in enumerate(names):
        p, a, l = s.last_model_outputs(i, len(pose))
        assert np.abs(p - CG[n + "/pose"]).max() < 1e-4 and np.abs(a - CG[n + "/affinity"]).max() < 1e-4


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "dense"])
def test_chunking_and_batch_independence(capi, CG, name):
    """A pose's score must not depend on batch size / chunking (poses are independent): the latency tiles of small
    launches and the zero-skipping K loops (first conv: per-tile quad lists; ReLU'd layers: per-MFMA tests in a fixed
    channel-major order) give bit-identical scores."""
    from gnina_amd import synth
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = oracle_maps(blob)
    rec_xyz, rec_smt, lig_smt, poses = synth.make_complex(5, synth.mapped_types(rmap[0]),
                                                          synth.mapped_types(lmap[0]), 1500, 24, 11)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    full = s.score_batch(poses, lig_smt)
    s.set_chunk(4)
    chunked = s.score_batch(poses, lig_smt)
    single = s.score_batch(poses[5:6], lig_smt)
    assert np.array_equal(full["pose"], chunked["pose"]) and np.array_equal(full["affinity"], chunked["affinity"])
    assert single["pose"][0] == full["pose"][5] and single["affinity"][0] == full["affinity"][5]
    # 605 poses: every layer runs its throughput tile (>= 512 workgroups), the 11-pose calls above their latency tiles
    s.set_chunk(1024)
    big = s.score_batch(np.concatenate([poses] * 55), lig_smt)
    for k in ("pose", "affinity"):
        assert np.array_equal(big[k].reshape(55, 11), np.tile(full[k], (55, 1)))
    # and against the oracle end to end
    grids = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap)[0] for b in (0, 5, 10)])
    with torch.no_grad():
        p, a, _ = cnn_ref.scores(blob, grids)
    assert np.abs(full["pose"][[0, 5, 10]] - p.numpy()).max() < 1e-4
    assert np.abs(full["affinity"][[0, 5, 10]] - a.numpy()).max() < 1e-4


def test_error_paths(capi):
    s = capi.Scorer(["default2017"])
    with pytest.raises(capi.MiGninaError):      # score before set_receptor (MI_ERR_STATE)
        s.score_batch(np.zeros((1, 2, 3), dtype=np.float32), np.array([2, 2]))
    with pytest.raises(capi.MiGninaError):      # unreadable model -> usage_error in the reference
        capi.Model("/nonexistent/model.mgw")
    s.set_receptor(np.zeros((1, 3), dtype=np.float32), np.array([2]))
    with pytest.raises(capi.MiGninaError):      # smina type out of range
        s.score_batch(np.zeros((1, 1, 3), dtype=np.float32), np.array([99]))


def test_ragged_batch_equals_per_ligand_calls(capi, CG):
    """Virtual-screening batches: poses of different ligands, padded to Lmax, in one call --
    bit-identical to scoring every ligand on its own."""
    from gnina_amd import synth
    names = ["crossdock_default2018", "crossdock_default2018_KD_4", "default2017"]
    base = "crossdock_default2018"
    rec_xyz, rec_smt = CG[base + "/rec_xyz"], CG[base + "/rec_smt"]
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    rng = np.random.RandomState(3)
    types = np.array([2, 3, 4, 5, 6, 8, 10, 12, 14, 1, 16], dtype=np.int32)   # includes polar H (untyped)
    Lmax, P = 48, 3
    xyz = np.zeros((0, Lmax, 3), dtype=np.float32)
    smt = np.zeros((0, Lmax), dtype=np.int32)
    single = {k: [] for k in ("pose", "affinity", "loss", "variance")}
    for L in (16, 48, 1, 33):
        lig_xyz, lig_smt = synth.make_ligand(rng, L, types)
        poses = synth.make_poses(rng, lig_xyz, P)
        out = s.score_batch(poses, lig_smt)
        for k in single:
            single[k].append(out[k])
        pad_xyz = np.full((P, Lmax, 3), 1e30, dtype=np.float32)     # padding coordinates must be ignored
        pad_xyz[:, :L] = poses
        pad_smt = np.full((P, Lmax), -1, dtype=np.int32)
        pad_smt[:, :L] = lig_smt
        xyz, smt = np.concatenate([xyz, pad_xyz]), np.concatenate([smt, pad_smt])
    out = s.score_ragged(xyz, smt)
    for k in single:
        assert np.array_equal(out[k], np.concatenate(single[k])), k
    bad = smt.copy()
    bad[0, 3] = -1                       # a hole in the middle of a ligand
    with pytest.raises(capi.MiGninaError):
        s.score_ragged(xyz, bad)


def test_full_size_batch_properties(capi, CG):
    """BASELINE config C2 at full size (1,024 poses): the oracle cannot run that many in seconds, so the batch
    is checked through size-independent properties -- run-to-run determinism, permutation equivariance,
    independence of the internal chunking, and agreement of a 16-pose sample with the oracle."""
    from gnina_amd import synth
    name = "default2017"
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = oracle_maps(blob)
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(rmap[0]))
    lig_xyz, lig_smt = synth.make_ligand(rng, 32, synth.mapped_types(lmap[0]))
    poses = synth.make_poses(np.random.RandomState(1000), lig_xyz, 1024)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    a = s.score_batch(poses, lig_smt)
    b = s.score_batch(poses, lig_smt)
    for k in ("pose", "affinity", "loss"):
        assert np.array_equal(a[k], b[k]), k                     # deterministic
    perm = np.random.RandomState(3).permutation(1024)
    c = s.score_batch(poses[perm], lig_smt)
    for k in ("pose", "affinity", "loss"):
        assert np.array_equal(c[k], a[k][perm]), k               # poses are independent of their neighbours
    s.set_chunk(37)
    d = s.score_batch(poses, lig_smt)
    for k in ("pose", "affinity", "loss"):
        assert np.array_equal(d[k], a[k]), k                     # ... and of the chunking / launch plan
    assert np.isfinite(a["pose"]).all() and (a["pose"] >= 0).all() and (a["pose"] <= 1).all()
    assert np.allclose(a["loss"], -np.log(np.maximum(a["pose"], 1e-30)), rtol=2e-4, atol=2e-6)  # CE(logp, 1) = -log p1
    sample = np.random.RandomState(5).choice(1024, 16, replace=False)
    grids = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, poses[i], lig_smt, rmap, lmap)[0] for i in sample])
    with torch.no_grad():
        p, aff, _ = cnn_ref.scores(blob, grids)
    assert np.abs(a["pose"][sample] - p.numpy()).max() < 1e-4
    assert np.abs(a["affinity"][sample] - aff.numpy()).max() < 1e-4


ALL = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens_all.npz"))


@pytest.mark.parametrize("name", [str(n) for n in ALL["names"]])
def test_every_builtin_model_matches_the_references_torchscript(capi, name):
    """All 64 built-in models (cnn_torch_scorer.cpp:24-64: default2017, the default2018 / dense families, every
    *_ensemble member, `fast`, `default1.0`), HIP vs the reference's own .pt outputs on the same atoms.  Blobs other
    than the seven committed ones are extracted from the reference at build time (gnina_amd/build.py)."""
    if not os.path.exists(os.path.join(WEIGHTS, name + ".mgw")):
        pytest.skip(f"{name}.mgw not extracted (python -m gnina_amd.build where the reference is present)")
    sig = f"sig{int(ALL[name + '/sig'])}"
    s = capi.Scorer([name])
    s.set_receptor(ALL[sig + "/rec_xyz"], ALL[sig + "/rec_smt"])
    out = s.score_batch(ALL[sig + "/poses"], ALL[sig + "/lig_smt"])
    assert np.abs(out["pose"] - ALL[name + "/pose"]).max() < 1e-4
    assert np.abs(out["affinity"] - ALL[name + "/affinity"]).max() < 1e-4
    assert np.abs(out["loss"] - ALL[name + "/loss"]).max() < 1e-3 * max(1.0, float(np.abs(ALL[name + "/loss"]).max()))


def _rot_matrix(q):
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def test_rotated_scoring_equals_scoring_the_rotated_complex(capi, CG):
    """TorchModel::forward(rotate = true): libmolgrid's Transform turns receptor and ligand about the grid centre
    before GridMaker::forward (torch_model.cpp:170-173).  mi_scorer_set_rotations must give what scoring the
    explicitly rotated atoms gives (grid, scores), and the atom gradients must come back in the unrotated frame."""
    name = "default2017"
    rec_xyz, rec_smt = CG[name + "/rec_xyz"], CG[name + "/rec_smt"]
    poses, lig_smt = CG[name + "/poses"], CG[name + "/lig_smt"]
    B = len(poses)
    rng = np.random.RandomState(5)
    q = rng.normal(size=(B, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = (1, 0, 0, 0)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    centers = poses.mean(1).astype(np.float32)        # lig.center() of the unrotated ligand
    s.set_rotations(q)
    out = s.score_batch(poses, lig_smt, centers=centers)
    plain = s.score_batch(poses, lig_smt, centers=centers)            # rotations are consumed by one call
    assert abs(out["pose"][0] - plain["pose"][0]) < 1e-6 and np.abs(out["pose"][1:] - plain["pose"][1:]).max() > 1e-5
    s.set_rotations(q)
    g_rot = s.score_grad(poses, lig_smt, centers=centers)
    s.set_rotations(q)
    grids_rot, _ = s.voxelize_batch(poses, lig_smt, centers=centers)    # the exported grid is rotated as well ...
    grids_plain, _ = s.voxelize_batch(poses, lig_smt, centers=centers)  # ... and the rotations are consumed by that call
    assert np.abs(grids_rot[0] - grids_plain[0]).max() < 1e-5 and np.abs(grids_rot[1:] - grids_plain[1:]).max() > 1e-2   # (q[0] = identity)
    for b in range(B):
        R = _rot_matrix(q[b])
        c = centers[b].astype(np.float64)
        s2 = capi.Scorer([name])
        s2.set_receptor(((rec_xyz - c) @ R.T + c).astype(np.float32), rec_smt)
        lig_r = ((poses[b] - c) @ R.T + c).astype(np.float32)
        o2 = s2.score_grad(lig_r[None], lig_smt, centers=centers[b:b + 1])
        g2, _ = s2.voxelize_batch(lig_r[None], lig_smt, centers=centers[b:b + 1])
        assert np.abs(grids_rot[b] - g2[0]).max() < 2e-4                 # (atoms rotated in fp32 on the device, fp64 here)
        assert abs(out["pose"][b] - o2["pose"][0]) < 2e-5 and abs(out["affinity"][b] - o2["affinity"][0]) < 2e-4
        back = o2["lig_grad"][0] @ R                                   # R^T g, row-vector form
        scale = max(1e-6, np.abs(back).max())
        assert np.abs(g_rot["lig_grad"][b] - back).max() < 2e-3 * scale
    with pytest.raises(capi.MiGninaError):
        s.set_rotations(q * 2)                                         # not unit length
    s.set_rotations(q[:2])
    with pytest.raises(capi.MiGninaError):
        s.score_batch(poses, lig_smt)                                  # pose count mismatch
    assert np.allclose(s.score_batch(poses, lig_smt, centers=centers)["pose"], plain["pose"], atol=1e-6)


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "dense"])
def test_zero_skipping_k_order_against_the_plain_order(capi, name, monkeypatch):
    """The per-MFMA zero test runs the K loop channel-major (and, for the Dense blocks, takes the BatchNorm shift out of
    the loop into a border-class bias table): same sums, re-associated.  MI_GNINA_NO_RELU_SKIP=1 is the plain tap-major
    path of round 1; the two must agree to rounding."""
    from gnina_amd import synth
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = oracle_maps(blob)
    rec_xyz, rec_smt, lig_smt, poses = synth.make_complex(7, synth.mapped_types(rmap[0]),
                                                          synth.mapped_types(lmap[0]), 1500, 24, 40)
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    new = s.score_batch(poses, lig_smt)
    with capi.option("MI_GNINA_NO_RELU_SKIP"):
        s0 = capi.Scorer([capi.Model(name)])   # a fresh model: the Dense-block plan reads the switch when it packs its weights
        s0.set_receptor(rec_xyz, rec_smt)
        old = s0.score_batch(poses, lig_smt)
    assert np.abs(new["pose"] - old["pose"]).max() < 5e-6
    assert np.abs(new["affinity"] - old["affinity"]).max() < 2e-5


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "dense"])
def test_scores_on_nearly_empty_grids(capi, name):
    """Edge cases of the zero-skipping paths: a ligand 60 A away from the receptor (only ligand channels populated: most
    channel quads of most tiles empty, whole staging windows of the voxelizer written as zeros), a ligand with no typed
    atom (receptor channels only), and a single typed atom next to the box corner."""
    from gnina_amd import synth
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    rmap, lmap = oracle_maps(blob)
    rng = np.random.RandomState(11)
    rec_xyz, rec_smt = synth.make_receptor(rng, 1200, synth.mapped_types(rmap[0]))
    lig_types = synth.mapped_types(lmap[0])
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    L = 12
    lig_smt = rng.choice(lig_types, L).astype(np.int32)
    base = rng.normal(0, 2.0, (L, 3)).astype(np.float32)
    far = base + np.array([60.0, -45.0, 30.0], dtype=np.float32)
    cases = [(far, lig_smt), (base, np.zeros(L, dtype=np.int32) + 0)]
    untyped = [t for t in range(28) if lmap[0][t] < 0]
    if untyped:
        cases[1] = (base, np.full(L, untyped[0], dtype=np.int32))
    one = base.copy()
    one_smt = np.full(L, untyped[0] if untyped else lig_smt[0], dtype=np.int32)
    one_smt[0] = lig_smt[0]
    cases.append((one, one_smt))
    for xyz, smt in cases:
        got = s.score_batch(xyz[None], smt)
        grid, _ = voxel.voxelize_pose(rec_xyz, rec_smt, xyz, smt, rmap, lmap)
        with torch.no_grad():
            p, a, _ = cnn_ref.scores(blob, grid[None])
        assert abs(float(got["pose"][0]) - float(p[0])) < 1e-4
        assert abs(float(got["affinity"][0]) - float(a[0])) < 1e-4
    # the same three poses inside one large launch (throughput tiles, sparsity probe has run): identical bits
    xyz3 = np.stack([c[0] for c in cases])
    big_smt = cases[0][1]
    alone = s.score_batch(xyz3[:1], big_smt)
    many = s.score_batch(np.concatenate([xyz3[:1]] * 600), big_smt)
    assert np.all(many["pose"] == alone["pose"][0]) and np.all(many["affinity"] == alone["affinity"][0])
