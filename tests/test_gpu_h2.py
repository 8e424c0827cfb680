"""The split-fp16 forward convolutions (gnina_amd/csrc/conv3d_h2.hip) against the fp32-MFMA kernels of the same layers.

Both are the parity path (MI_PRECISION_FP32 picks the split kernels where a layer has that plan, MI_PRECISION_FP32_MFMA
forces fp32 MFMA): every product of the split path is exact and the accumulation is fp32, so the scores may differ by
rounding noise only.  The bar here is 1e-5, a tenth of the parity bar against the reference; the measured differences are
printed.  Parity against the reference's own outputs is what tests/test_gpu_parity.py checks -- on the default path,
i.e. on these kernels.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODELS = ["default2017", "crossdock_default2018", "crossdock_default2018_KD_4", "dense", "dense_1_3", "dense_1_3_PT_KD_3"]
TOL = 2e-5  # (both paths sit a few 1e-6 from the float64 forward of the same operands; measured differences are printed)


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def CG(golden_dir):
    return np.load(os.path.join(golden_dir, "cnn_goldens.npz"))


def _both(capi, models, rec_xyz, rec_smt, poses, lig_smt):
    s = capi.Scorer(models)
    s.set_receptor(rec_xyz, rec_smt)
    s.set_precision("fp32_mfma")
    ref = s.score_batch(poses, lig_smt)
    s.set_precision("fp32")
    new = s.score_batch(poses, lig_smt)
    return ref, new


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("B", [4, 96])
def test_split_fp16_scores_equal_fp32_mfma_scores(capi, CG, name, B):
    from gnina_amd import synth
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    if B > len(poses):  # 4 poses take the latency tiles, 96 the throughput tiles
        poses = np.concatenate([poses, synth.make_poses(np.random.RandomState(5), poses[0] - poses[0].mean(0), B - len(poses))])
    ref, new = _both(capi, [name], rec_xyz, rec_smt, poses, lig_smt)
    dp, da = np.abs(new["pose"] - ref["pose"]).max(), np.abs(new["affinity"] - ref["affinity"]).max()
    print(f"{name} B={B}: split-fp16 vs fp32 MFMA max|d pose| {dp:.2e} max|d affinity| {da:.2e}")
    assert np.isfinite(new["pose"]).all() and np.isfinite(new["affinity"]).all()
    assert dp <= TOL and da <= TOL
    if B == 4:  # and the default path meets the reference's own outputs
        assert np.abs(new["pose"][:4] - CG[name + "/pose"]).max() < 1e-4
        assert np.abs(new["affinity"][:4] - CG[name + "/affinity"]).max() < 1e-4 * max(1.0, np.abs(CG[name + "/affinity"]).max())


def test_split_fp16_is_independent_of_the_batch_size_and_tiling(capi, CG):
    """the latency tiles (small launches) and the throughput tiles walk K in the same order: same bits"""
    from gnina_amd import synth
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    many = np.concatenate([poses, synth.make_poses(np.random.RandomState(7), poses[0] - poses[0].mean(0), 92)])
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    big = s.score_batch(many, lig_smt)
    one = s.score_batch(many[:1], lig_smt)
    assert one["pose"][0] == big["pose"][0] and one["affinity"][0] == big["affinity"][0]


def test_split_fp16_on_nearly_empty_and_crowded_grids(capi, CG):
    """zero skipping on an (almost) empty grid; many atoms on one spot (large density values)"""
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    far = poses.copy()
    far[0] += 200.0
    crowded = (np.repeat(poses[:, :1], poses.shape[1], axis=1) +
               0.01 * np.arange(poses.shape[1], dtype=np.float32)[None, :, None]).astype(np.float32)
    for batch in (far, crowded):
        ref, new = _both(capi, [name], rec_xyz, rec_smt, batch, lig_smt)
        assert np.abs(new["pose"] - ref["pose"]).max() <= TOL
        assert np.abs(new["affinity"] - ref["affinity"]).max() <= TOL * max(1.0, float(np.abs(ref["affinity"]).max()))


def test_fused_1x1_conv_gives_the_bits_of_two_launches(capi, CG, monkeypatch):
    """Default2018's "3x3x3 conv -> ReLU -> 1x1x1 conv" pairs run as one split-fp16 kernel (the ReLU'd tile is split and
    laid down in LDS, never in HBM); as two launches (MI_GNINA_H2_NO_FUSE1X1, read when a model is loaded) the operands and
    the MFMA order are the same, so the scores must be too -- which is what lets the gradient program, which needs the
    intermediate activation, run the pair separately and still score a pose like the forward program."""
    from gnina_amd import synth
    name = "crossdock_default2018"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    many = np.concatenate([poses, synth.make_poses(np.random.RandomState(11), poses[0] - poses[0].mean(0), 60)])
    out = []
    for no_fuse in (False, True):
        capi.set_option("MI_GNINA_H2_NO_FUSE1X1", "1" if no_fuse else None)   # (read when a model is loaded)
        s = capi.Scorer([capi.Model(name)])
        s.set_receptor(rec_xyz, rec_smt)
        out.append(s.score_batch(many, lig_smt))
        capi.set_option("MI_GNINA_H2_NO_FUSE1X1", None)
    assert np.array_equal(out[0]["pose"], out[1]["pose"]) and np.array_equal(out[0]["affinity"], out[1]["affinity"])
    assert np.abs(out[0]["pose"][:4] - CG[name + "/pose"]).max() < 1e-4


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018"])
def test_two_poses_per_workgroup_and_an_odd_batch(capi, CG, name):
    """the throughput tiles of the first conv take two consecutive poses per workgroup on one copy of the weights in LDS
    (conv3d_h2_kernel NP = 2); an odd batch leaves the last workgroups one pose.  Same K order: every pose scores the bits
    it scores alone, and MI_GNINA_H2_WLDS=0 (weights from L1 / L2, one pose per workgroup) gives the same bits."""
    from gnina_amd import synth
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    many = np.concatenate([poses, synth.make_poses(np.random.RandomState(13), poses[0] - poses[0].mean(0), 93)])   # 97 poses
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    big = s.score_batch(many, lig_smt)
    even = s.score_batch(many[:96], lig_smt)
    assert np.array_equal(big["pose"][:96], even["pose"]) and np.array_equal(big["affinity"][:96], even["affinity"])
    for b in (0, 95, 96):
        one = s.score_batch(many[b:b + 1], lig_smt)
        assert one["pose"][0] == big["pose"][b] and one["affinity"][0] == big["affinity"][b], b
    with capi.option("MI_GNINA_H2_WLDS", 0):
        plain = s.score_batch(many, lig_smt)
    assert np.array_equal(plain["pose"], big["pose"]) and np.array_equal(plain["affinity"], big["affinity"])


@pytest.mark.parametrize("name", ["default2017", "dense"])
def test_stationary_weights_first_conv_gives_the_same_bits(capi, CG, name):
    """conv3d_h2_ws_kernel (conv3d_h2_ws.hip, round 6, opt-in MI_GNINA_H2_WS = ring size): one persistent workgroup per CU,
    a chunk's weights DMA'd once per 8 poses, the accumulators of 8 poses resident, halo tiles through a ring with hand-counted
    vmcnt.  Same tile, K order and MFMA order per accumulator as conv3d_h2_kernel: same bits, for batches that are and are
    not multiples of 8, for every ring size, for the pooled (default2017) and the un-pooled split-output (Dense) first conv."""
    from gnina_amd import synth
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    many = np.concatenate([poses, synth.make_poses(np.random.RandomState(17), poses[0] - poses[0].mean(0), 63)])   # 67 poses
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    ref = s.score_batch(many, lig_smt)
    assert np.abs(ref["pose"][:len(poses)] - CG[name + "/pose"]).max() < 1e-4
    for ring in (2, 4, 5):
        with capi.option("MI_GNINA_H2_WS", ring):
            got = s.score_batch(many, lig_smt)
            part = s.score_batch(many[:40], lig_smt)
        assert np.array_equal(got["pose"], ref["pose"]) and np.array_equal(got["affinity"], ref["affinity"]), ring
        assert np.array_equal(part["pose"], ref["pose"][:40]) and np.array_equal(part["affinity"], ref["affinity"][:40]), ring


def test_dense_per_pose_variants_give_the_same_bits(capi, CG):
    """What a per-pose call of a Dense model runs differently since round 6 (also: chunk groups in the 1x1x1 transitions, the
    whole-grid max pool fused with the heads) -- the K chunks of a block layer DMA'd
    in groups (conv3d_h2_d16_kernel<.., GRP>, conv3d_h2_dense.hip) and the 6^3 layers' weights through LDS, a chunk ahead
    (conv3d_h2_16_kernel<.., WL>; conv3d_h2_16_ring_kernel with both operands through a ring for launches of few small
    workgroups; conv3d_h2.hip) -- feed the same operands to the same MFMAs in the same order: B = 1, B = 3
    and a batch score the same bits with the variants on (default), off, and with the weight buffers on every 16-wide layer."""
    from gnina_amd import synth
    name = "dense"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    many = np.concatenate([poses, synth.make_poses(np.random.RandomState(23), poses[0] - poses[0].mean(0), 29)])   # 33 poses
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    ref = s.score_batch(many, lig_smt)
    assert np.abs(ref["pose"][:len(poses)] - CG[name + "/pose"]).max() < 1e-4

    def check(tag):
        one = [s.score_batch(many[k:k + 1], lig_smt) for k in range(4)]
        three = s.score_batch(many[4:7], lig_smt)
        full = s.score_batch(many, lig_smt)
        for k in range(4):
            assert one[k]["pose"][0] == ref["pose"][k] and one[k]["affinity"][0] == ref["affinity"][k], (tag, k)
        assert np.array_equal(three["pose"], ref["pose"][4:7]) and np.array_equal(three["affinity"], ref["affinity"][4:7]), tag
        assert np.array_equal(full["pose"], ref["pose"]) and np.array_equal(full["affinity"], ref["affinity"]), tag

    check("default")
    with capi.option("MI_GNINA_D16_DBG", 128):
        check("no chunk groups")
    with capi.option("MI_GNINA_H16_WLDS", 0):
        check("weights from L2")
    with capi.option("MI_GNINA_H16_WLDS", 2):
        check("weights through LDS wherever they fit")
    with capi.option("MI_GNINA_H16_WLDS", 5):
        check("no operand ring for the launches of few small workgroups")
    with capi.option("MI_GNINA_NO_GMAX_FUSE", 1):
        check("whole-grid max pool and heads as two launches (one for small calls: gmax_heads_kernel)")
    with capi.option("MI_GNINA_K1S_DBG", 128):
        check("the 1x1x1 transitions' chunks one DMA round trip each")
    with capi.option("MI_GNINA_H16_WLDS", 9):
        check("the ring kernel without its producer / consumer form (conv3d_h2_16_pc_kernel)")
