"""Two scorers of one device driven from two host threads (the C ABI allows it; gnina itself scores under DLScorer::mtx,
gninasrc/lib/dl_scorer.h:26, cnn_torch_scorer.cpp:106): every call must give the bits the scorer gives alone.

Round 5 found that it did not -- with a Dense model on the second thread ~5 % of the B = 1 calls deviated (up to 3e-2 in the
affinity): kernels of two hardware queues running side by side change what the voxelizer accumulates.  The library now holds
a per-device lock for the duration of a host-output scoring call (engine.cpp device_call_lock), which is also the reference's
behaviour.  The ensemble path that ran an ensemble's models on their own streams ("lanes") is opt-in for the same reason."""
import os
import threading

import numpy as np
import pytest

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
    (lanes; every voxel group is voxelized before the first lane starts).  Same bits every time, the goldens' scores, and
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
