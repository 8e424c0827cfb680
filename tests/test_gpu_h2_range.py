"""The split-fp16 forward path at the edges of the fp16 range (gnina_amd/csrc/conv3d_h2.hip, include/mi_gnina.h
MI_PRECISION_FP32).

The reference runs any --cnn_model in fp32 (gninasrc/lib/torch_model.cpp:49-118,185); the default path of this library
writes every activation as two fp16 halves, so
  * an activation beyond +-65504 cannot be represented: every split-fp16 kernel (and the voxelizer feeding one) raises the
    scorer's range flag when it produces or consumes such a value, and a call that ends with the flag raised is repeated
    on the fp32-MFMA kernels -- the scores a caller sees are the fp32 ones;
  * an activation below 2^-14 has a subnormal high half (absolute error <= 2^-25): harmless at the parity bar.
Both ends are driven here against the float64 forward of the CPU oracle (oracle/cnn_ref.py) at the parity bar, 1e-4.
"""
import os
import struct

import numpy as np
import pytest
import torch

from oracle import cnn_ref

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


def scaled_blob(name, k, tmp_path):
    """the model `name` with its FIRST convolution (weights and bias) multiplied by k and its fully connected weights by
    1 / k (k a power of two: exact): every activation behind the first conv is ~k times larger, the scores stay O(1)"""
    raw = bytearray(open(os.path.join(WEIGHTS, name + ".mgw"), "rb").read())
    blob = cnn_ref.Blob(bytes(raw))
    (hl,) = struct.unpack("<I", raw[8:12])
    off = 12 + hl
    off += (-off) % 64
    data = np.frombuffer(raw, dtype="<f4", offset=off).copy()
    conv = next(t for t in blob.ops if t[0] == "conv")
    kk, cin, cout = int(conv[1]), int(conv[4]), int(conv[5])
    w_off, b_off = int(conv[8]), int(conv[9])
    data[w_off:w_off + kk ** 3 * cin * cout] *= np.float32(k)
    data[b_off:b_off + cout] *= np.float32(k)
    fc = next(t for t in blob.ops if t[0] == "fc")
    n_in, fw_off = int(fc[2]), int(fc[3])
    data[fw_off:fw_off + 3 * n_in] *= np.float32(1.0 / k)
    raw[off:] = data.tobytes()
    path = os.path.join(str(tmp_path), f"{name}_x{k:g}.mgw")
    with open(path, "wb") as f:
        f.write(bytes(raw))
    return path


def oracle_scores(path, grids):
    blob = cnn_ref.Blob(path)
    pose, aff, _ = cnn_ref.scores(blob, torch.from_numpy(grids), dtype=torch.float64)
    return pose.numpy(), aff.numpy()


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018"])
def test_activations_beyond_the_fp16_range_fall_back_to_fp32_mfma(capi, CG, name, tmp_path):
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = scaled_blob(name, 2.0 ** 18, tmp_path)  # first-layer activations of ~0.5 .. 5 become 1e5 .. 1e6
    s = capi.Scorer([capi.Model(path)])
    s.set_receptor(rec_xyz, rec_smt)
    grids, _ = s.voxelize_batch(poses, lig_smt)
    want_pose, want_aff = oracle_scores(path, grids)
    assert s.h2_fallbacks() == 0
    got = s.score_batch(poses, lig_smt)                      # default precision: split-fp16 kernels, flag, repeat
    assert s.h2_fallbacks() == 1, "the split-fp16 kernels did not report activations beyond 65504"
    assert np.isfinite(got["pose"]).all() and np.isfinite(got["affinity"]).all()
    scale = max(1.0, float(np.abs(want_aff).max()))
    assert np.abs(got["pose"] - want_pose).max() < 1e-4
    assert np.abs(got["affinity"] - want_aff).max() < 1e-4 * scale
    s.set_precision("fp32_mfma")                             # ... and they are the fp32-MFMA path's bits
    ref = s.score_batch(poses, lig_smt)
    assert np.array_equal(ref["pose"], got["pose"]) and np.array_equal(ref["affinity"], got["affinity"])
    assert s.h2_fallbacks() == 1
    # gradient calls take the same route
    s.set_precision("fp32")
    g = s.score_grad(poses, lig_smt)
    assert s.h2_fallbacks() == 2
    assert np.abs(g["pose"] - got["pose"]).max() <= 1e-5 and np.isfinite(g["lig_grad"]).all()
    # a device-output call cannot repeat itself: mi_scorer_synchronize reports MI_ERR_RANGE
    d_lig = torch.from_numpy(poses).cuda()
    d_out = torch.empty(4, len(poses), dtype=torch.float32, device="cuda")
    s.score_batch_device(d_lig.data_ptr(), lig_smt, len(poses), poses.shape[1], d_out[0].data_ptr(), d_out[1].data_ptr(),
                         d_out[2].data_ptr(), d_out[3].data_ptr())
    with pytest.raises(capi.MiGninaError):
        s.synchronize()
    s.synchronize()                                          # the flag is consumed
    # the unscaled model never raises it
    s0 = capi.Scorer([name])
    s0.set_receptor(rec_xyz, rec_smt)
    s0.score_batch(poses, lig_smt)
    assert s0.h2_fallbacks() == 0


@pytest.mark.parametrize("name", ["default2017", "crossdock_default2018", "dense"])
def test_tiny_activations_keep_the_parity_bar(capi, name):
    """voxel grids whose values sweep [1e-8, 6e-5] -- subnormal fp16 high halves, low halves that vanish -- through the
    split-fp16 kernels (mi_model_forward_grids pools the grid and hands the fp32 tensor to the same layer program)"""
    m = capi.Model(name)
    s = capi.Scorer([m])
    C, N = m.n_channels, m.grid_points
    rng = np.random.RandomState(3)
    B = 3
    grids = np.exp(rng.uniform(np.log(1e-8), np.log(6e-5), (B, C, N, N, N))).astype(np.float32)
    grids[rng.uniform(size=grids.shape) < 0.5] = 0.0
    got_pose, got_aff, _ = s.forward_grids(grids)
    want_pose, want_aff = oracle_scores(os.path.join(WEIGHTS, name + ".mgw"), grids)
    assert s.h2_fallbacks() == 0
    assert np.abs(got_pose - want_pose).max() < 1e-4
    assert np.abs(got_aff - want_aff).max() < 1e-4 * max(1.0, float(np.abs(want_aff).max()))


def test_values_at_the_edge_of_the_range_do_not_raise_the_flag(capi, CG, tmp_path):
    """the flag is about the fp16 RANGE, not about large values: a model scaled so that its activations reach a few 1e3
    stays on the split-fp16 kernels and meets the float64 forward"""
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = scaled_blob(name, 2.0 ** 6, tmp_path)
    s = capi.Scorer([capi.Model(path)])
    s.set_receptor(rec_xyz, rec_smt)
    grids, _ = s.voxelize_batch(poses, lig_smt)
    want_pose, want_aff = oracle_scores(path, grids)
    got = s.score_batch(poses, lig_smt)
    assert s.h2_fallbacks() == 0
    assert np.abs(got["pose"] - want_pose).max() < 1e-4
    assert np.abs(got["affinity"] - want_aff).max() < 1e-4 * max(1.0, float(np.abs(want_aff).max()))


def test_pool_repeats_a_device_resident_shard_on_fp32_mfma(capi, CG, tmp_path, allow_duplicate_devices):
    """mi_pool's device-resident path cannot see the range flag until mi_scorer_synchronize returns MI_ERR_RANGE: the
    worker then scores its shard again under MI_PRECISION_FP32_MFMA (pool.cpp score_resident) -- the caller gets the
    fp32 bits, from a pool of one device and from a sharded pool alike."""
    name = "default2017"
    rec_xyz, rec_smt, lig_smt, poses = (CG[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    path = scaled_blob(name, 2.0 ** 18, tmp_path)
    s = capi.Scorer([capi.Model(path)])
    s.set_receptor(rec_xyz, rec_smt)
    s.set_precision("fp32_mfma")
    want = s.score_batch(poses, lig_smt)
    d_lig = torch.from_numpy(poses).cuda()
    for devices in ([0], [0, 0]):
        pool = capi.Pool([path], devices)
        pool.set_receptor(rec_xyz, rec_smt)
        d_out = torch.zeros(4, len(poses), dtype=torch.float32, device="cuda")
        pool.score_batch_device(d_lig.data_ptr(), lig_smt, len(poses), poses.shape[1], d_out[0].data_ptr(), d_out[1].data_ptr(),
                                d_out[2].data_ptr(), d_out[3].data_ptr())
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert np.array_equal(got[0], want["pose"]) and np.array_equal(got[1], want["affinity"]), devices
