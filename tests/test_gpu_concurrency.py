"""Several scorers of one device driven from several host threads at the same time -- how gnina drives the seam: fresh_copy()
hands every worker thread / Monte-Carlo task a new CNNTorchScorer with a mutex of its own (gninasrc/lib/cnn_torch_scorer.h:54,
dl_scorer.h:43-44, main.cpp:1436-1438, parallel_mc.cpp:145-146).  Every call must give the bits the scorer gives alone.

Round 5 found that it did not (~5 % of B = 1 calls next to a Dense model, up to 6e-2 in the affinity) and serialised the calls
per device.  Round 6 found the cause -- voxelize_tiles' packed-fp32 instructions go wrong in the upper half of a wavefront
while the Dense family's f16-MFMA K loops share the SIMD (tools/experiments/vox_stress.py) --, took
those instructions out of the voxelizer and took the lock away: these tests run WITHOUT any serialisation, host-output and
device-output calls, and with an ensemble's models on their own streams (lanes: on by default for calls of <= 8 poses)."""
import os
import threading

import numpy as np
import pytest
import torch  # (before the library initialises HIP: a torch imported afterwards finds no device)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.mark.parametrize("pair", [("dense_1_3", "crossdock_default2018_KD_4"), ("dense_1_3", "dense_1_3_PT_KD_3")])
def test_two_threads_give_the_single_thread_bits(capi, pair):
    G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
    base = "dense_1_3"
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    N = 200

    def loop(s, out):
        for rep in range(N):
            b = rep % len(poses)
            r = s.score_batch(poses[b:b + 1], lig_smt)
            out.append((float(r["pose"][0]), float(r["affinity"][0])))

    scorers, refs = [], []
    for n in pair:
        s = capi.Scorer([n])
        s.set_receptor(rec_xyz, rec_smt)
        scorers.append(s)
        o = []
        loop(s, o)
        refs.append(np.array(o))
        assert np.abs(refs[-1][:len(poses), 0] - G[n + "/pose"]).max() < 1e-4
    outs = [[] for _ in scorers]
    th = [threading.Thread(target=loop, args=(s, o)) for s, o in zip(scorers, outs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for n, o, r in zip(pair, outs, refs):
        assert np.array_equal(np.array(o), r), (n, int((np.abs(np.array(o) - r).max(axis=1) > 0).sum()), "of", N, "calls deviate")


def test_default_ensemble_small_calls_are_reproducible(capi):
    """gnina's default ensemble at B = 1 (DLScorer::score as gnina calls it): the models' programs run on their own streams
    (lanes, behind the voxelization of the ensemble's groups).  Same bits every time, the goldens' scores, and
    the bits of the one-stream call (MI_GNINA_NO_LANES=1), per model."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
    names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    want_aff = np.mean([G[n + "/affinity"] for n in names], axis=0)
    first = None
    for rep in range(15):
        got = np.array([[float(x[0]) for x in (r["pose"], r["affinity"])] for r in (s.score_batch(poses[b:b + 1], lig_smt) for b in range(len(poses)))])
        if first is None:
            first = got
            assert np.abs(got[:, 1] - want_aff).max() < 1e-4
        assert np.array_equal(got, first), rep
    # per model, against the same calls on one stream
    def per_model(sc, reps):
        out = []
        for rep in range(reps):
            b = rep % len(poses)
            sc.score_batch(poses[b:b + 1], lig_smt)
            out.append([[float(x[0]) for x in sc.last_model_outputs(m, 1)[:2]] for m in range(len(names))])
        return np.array(out)
    lanes = per_model(s, 150)
    with capi.option("MI_GNINA_NO_LANES", "1"):
        s1 = capi.Scorer(names)
        s1.set_receptor(rec_xyz, rec_smt)
        serial = per_model(s1, 150)
    assert np.array_equal(lanes, serial), int((np.abs(lanes - serial).max(axis=(1, 2)) > 0).sum())


def test_default_ensemble_gradient_calls_on_lanes(capi):
    """Gradient calls (score + d loss / d atoms: CNN refinement, torch_model.cpp:197-221) of gnina's default ensemble at
    B = 1 .. 3 run every model -- forward program, backward pass, voxel backward -- on a stream and in a buffer set of its own,
    the per-model atom gradients added in model order at the end: the additions of the one-stream call.  Same scores and the
    same gradient bits as with MI_GNINA_NO_GRAD_LANES=1, every time, and a scoring call in between does not disturb them."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
    names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))

    def run(sc, reps):
        out = []
        for rep in range(reps):
            b, n = rep % len(poses), 1 + rep % 3
            batch = np.concatenate([poses, poses])[b:b + n]
            r = sc.score_grad(batch, lig_smt)
            if rep % 5 == 0:
                sc.score_batch(batch[:1], lig_smt)
            out.append((r["pose"].copy(), r["affinity"].copy(), r["loss"].copy(), r["lig_grad"].copy()))
        return out

    with capi.option("MI_GNINA_NO_GRAD_LANES", "1"):
        s1 = capi.Scorer(names)
        s1.set_receptor(rec_xyz, rec_smt)
        serial = run(s1, 40)
        assert not s1.last_call_on_lanes() if hasattr(s1, "last_call_on_lanes") else True
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    lanes = run(s, 40)
    again = run(s, 40)
    for k, (a, b, c) in enumerate(zip(serial, lanes, again)):
        for x, y, z in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(y, z), k
    assert np.isfinite(lanes[0][3]).all() and np.abs(lanes[0][3]).max() > 0


def test_two_threads_device_output_calls(capi):
    """MI_LIG_ON_DEVICE | MI_OUT_ON_DEVICE calls (what the pools and a device-resident caller make) return after they enqueue:
    two threads then have voxelizers and conv kernels of two scorers in flight side by side for the whole run."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
    pair = ("crossdock_default2018_KD_4", "dense_1_3")
    rec_xyz, rec_smt, lig_smt, poses = (G[f"dense_1_3/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    N, P, L = 300, len(poses), poses.shape[1]
    dev = torch.device("cuda:0")
    d_lig = torch.from_numpy(np.ascontiguousarray(poses, np.float32)).to(dev)

    def loop(s, d_out):  # call k scores pose k % P into row k of d_out[0..3]
        for k in range(N):
            b = k % P
            s.score_batch_device(d_lig[b].data_ptr(), lig_smt, 1, L, d_out[0][k:].data_ptr(), d_out[1][k:].data_ptr(),
                                 d_out[2][k:].data_ptr(), d_out[3][k:].data_ptr())
        s.synchronize()

    scorers, refs, outs = [], [], []
    for n in pair:
        s = capi.Scorer([n])
        s.set_receptor(rec_xyz, rec_smt)
        scorers.append(s)
        d_ref = torch.zeros(4, N, device=dev)
        loop(s, d_ref)
        refs.append(d_ref.cpu().numpy())
        assert np.abs(refs[-1][0, :P] - G[n + "/pose"]).max() < 1e-4
        outs.append(torch.zeros(4, N, device=dev))
    torch.cuda.synchronize()
    th = [threading.Thread(target=loop, args=(s, o)) for s, o in zip(scorers, outs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    for n, o, r in zip(pair, outs, refs):
        got = o.cpu().numpy()
        assert np.array_equal(got[:2], r[:2]), (n, int((np.abs(got[:2] - r[:2]).max(axis=0) > 0).sum()), "of", N, "calls deviate")


def test_four_threads_default_ensemble(capi):
    """Four fresh copies of gnina's default ensemble scoring B = 1 poses from four threads (main.cpp:1436-1438): their lanes
    share the device's lane streams, their voxelizers run next to each other's Dense conv kernels; same bits as one thread."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
    names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
    rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
    N = 120

    def loop(s, out):
        for rep in range(N):
            b = rep % len(poses)
            r = s.score_batch(poses[b:b + 1], lig_smt)
            out.append((float(r["pose"][0]), float(r["affinity"][0])))

    scorers = []
    for _ in range(4):
        s = capi.Scorer(names)
        s.set_receptor(rec_xyz, rec_smt)
        scorers.append(s)
    ref = []
    loop(scorers[0], ref)
    ref = np.array(ref)
    outs = [[] for _ in scorers]
    th = [threading.Thread(target=loop, args=(s, o)) for s, o in zip(scorers, outs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i, o in enumerate(outs):
        assert np.array_equal(np.array(o), ref), (i, int((np.abs(np.array(o) - ref).max(axis=1) > 0).sum()), "of", N, "calls deviate")
