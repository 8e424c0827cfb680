"""Row a10: the CNN inside the optimisation loop.  HIP: mi_cnn_eval_batch (non_cache_cnn::eval / eval_deriv,
non_cache_cnn.cpp:33-54,79-169) and mi_cnn_refine_batch (refine_structure on it, main.cpp:131-171).
Oracle: oracle/cnn_refine.py (voxelizer + CNN autograd + torsion-tree fold + the same bfgs<>)."""
import os

import numpy as np
import pytest

from oracle import cnn_ref, cnn_refine, vina as ovina
from gnina_amd import vina_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WEIGHTS = os.path.join(ROOT, "gnina_amd", "weights")


@pytest.fixture(scope="module")
def setup():
    from gnina_amd import capi
    capi.init(0)
    sc = vina_scene.build(seed=3)
    lig = sc["lig"]
    lig["smt"] = lig["smt"].copy()
    lig["smt"][[5, 20]] = 1          # two polar hydrogens: go to the CNN untyped, get zero force
    v = capi.Vina()
    v.set_ligand(lig)
    olig = ovina.LigandHandle(lig)
    return capi, sc, lig, v, olig


def random_confs(lig, rng, n, trans=0.6, rot=0.25, tors=0.5):
    """conf0 perturbed: translation U(-trans, trans), a rotation of up to `rot` rad, torsion offsets"""
    out = []
    for _ in range(n):
        c = lig["conf0"].astype(np.float32).copy()
        c[:3] += rng.uniform(-trans, trans, 3)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(-rot, rot)
        r = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
        q = c[3:7]
        c[3:7] = [r[0] * q[0] - r[1] * q[1] - r[2] * q[2] - r[3] * q[3],
                  r[0] * q[1] + r[1] * q[0] + r[2] * q[3] - r[3] * q[2],
                  r[0] * q[2] - r[1] * q[3] + r[2] * q[0] + r[3] * q[1],
                  r[0] * q[3] + r[1] * q[2] - r[2] * q[1] + r[3] * q[0]]
        c[7:] += rng.uniform(-tors, tors, len(c) - 7)
        out.append(c)
    return np.stack(out).astype(np.float32)


def heavy_center(coords, smt):
    c = np.zeros(3, dtype=np.float32)
    for i in range(len(smt)):
        if smt[i] > 1:
            c = (c + coords[i]).astype(np.float32)
    return c / np.float32((smt > 1).sum())


@pytest.mark.parametrize("names", [["crossdock_default2018"], ["default2017", "crossdock_default2018_KD_4"]])
def test_cnn_eval_deriv_matches_oracle(setup, names):
    capi, sc, lig, v, olig = setup
    s = capi.Scorer(names)
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    rng = np.random.RandomState(11)
    confs = random_confs(lig, rng, 3)
    confs[2, :3] += [9.0, 0.0, -7.5]   # far enough to leave both the search box and the CNN cube
    coords = v.coords_batch(confs)
    for b in range(3):
        assert np.abs(coords[b] - ovina.set_conf(olig, confs[b])[0]).max() < 2e-5
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    cen = np.stack([heavy_center(ovina.set_conf(olig, lig["conf0"])[0], lig["smt"])] * 3)
    box = capi.CnnBox.make(23.5, lo, hi, slope=10.0)
    e, ch = v.cnn_eval_batch(s, confs, box, cen, deriv=True)
    e0, _ = v.cnn_eval_batch(s, confs, box, cen, deriv=False)
    assert np.abs(e - e0).max() < 1e-4 * np.abs(e).max()
    blobs = [cnn_ref.Blob(os.path.join(WEIGHTS, n + ".mgw")) for n in names]
    nc = cnn_refine.NonCacheCnn(blobs, sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), 23.5)
    nc.cnn_center = cen[0]
    nc.slope = 10.0
    for b in range(3):
        eo, cho = nc.eval_deriv(confs[b])
        assert abs(e[b] - eo) < 2e-4 * max(1.0, abs(eo)), (b, e[b], eo)
        assert np.abs(ch[b] - cho).max() < 3e-3 * max(np.abs(cho).max(), 1e-3), (b, ch[b], cho)
        assert abs(nc.eval(confs[b]) - eo) < 1e-4 * max(1.0, abs(eo))
    assert e[2] > e[0] + 10.0          # the displaced pose pays the out-of-box penalties


def test_cnn_refine_first_iterations_match_oracle(setup):
    capi, sc, lig, v, olig = setup
    name = "crossdock_default2018"
    s = capi.Scorer([name])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    rng = np.random.RandomState(5)
    confs = random_confs(lig, rng, 3, trans=0.4, rot=0.15, tors=0.3)
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    box = capi.CnnBox.make(23.5, lo, hi)
    e, out, tries, evals = v.cnn_refine_batch(s, confs, box, max_iters=2)
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    close = 0
    for b in range(3):
        nc = cnn_refine.NonCacheCnn([blob], sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), 23.5)
        eo, co, to = cnn_refine.refine_structure(nc, confs[b], 2)
        assert to == tries[b] == 1
        if abs(e[b] - eo) < 1e-3 * max(1.0, abs(eo)) and np.abs(out[b] - co).max() < 5e-3 and nc.evals == evals[b]:
            close += 1
    assert close >= 2


def test_cnn_refine_lowers_the_loss(setup):
    """--cnn_scoring refinement must end at a lower CNN loss than it started from (what test_min.py /
    test_cnn.py:56-59 check qualitatively), stay inside the box, and be reproducible."""
    capi, sc, lig, v, olig = setup
    names = ["crossdock_default2018", "dense_1_3"]
    s = capi.Scorer(names)
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    rng = np.random.RandomState(7)
    confs = random_confs(lig, rng, 16, trans=0.5, rot=0.2, tors=0.4)
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    box = capi.CnnBox.make(23.5, lo, hi)
    start, _ = v.cnn_eval_batch(s, confs, box, None, deriv=False)
    e, out, tries, evals = v.cnn_refine_batch(s, confs, box)
    # (a pose the search cannot improve keeps its first energy -- computed by the gradient program, which for the Dense
    # family is not the forward program bit for bit any more: BatchNorm folded into the weights there, applied while
    # staging here; the two agree to ~1e-6 relative)
    assert (e <= start + 2e-5 * np.maximum(1.0, np.abs(start))).all() and (e < start - 1e-3).sum() >= 12
    assert (tries == 1).all() and (evals >= 2).all()
    after, _ = v.cnn_eval_batch(s, out, box, None, deriv=False)
    assert np.abs(after - e).max() < 1e-4
    e2, out2, _, _ = v.cnn_refine_batch(s, confs, box)
    assert np.array_equal(e, e2) and np.array_equal(out, out2)


@pytest.mark.parametrize("mix_force,mix_energy", [(True, False), (True, True), (False, True)])
def test_cnn_eval_deriv_with_empirical_mix(setup, mix_force, mix_energy):
    """cnn_options::mix_emp_force / mix_emp_energy / empirical_weight (non_cache_cnn.cpp:113-166)."""
    capi, sc, lig, v, olig = setup
    name = "crossdock_default2018"
    s = capi.Scorer([name])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    rng = np.random.RandomState(21)
    confs = random_confs(lig, rng, 2)
    confs[1, :3] += [6.0, 0.0, 0.0]                 # partly outside the search box
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    cen = np.stack([heavy_center(ovina.set_conf(olig, lig["conf0"])[0], lig["smt"])] * 2)
    wgt = 0.7
    box = capi.CnnBox.make(23.5, lo, hi, slope=10.0, mix_emp_force=mix_force, mix_emp_energy=mix_energy,
                           empirical_weight=wgt, v=1000.0)
    e, ch = v.cnn_eval_batch(s, confs, box, cen, deriv=True)
    plain = capi.CnnBox.make(23.5, lo, hi, slope=10.0)
    e_plain, ch_plain = v.cnn_eval_batch(s, confs, plain, cen, deriv=True)
    e0, _ = v.cnn_eval_batch(s, confs, box, cen, deriv=False)
    assert np.abs(e0 - e_plain).max() < 1e-4 * np.abs(e_plain).max()     # ::eval ignores the mix options
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    nc = cnn_refine.NonCacheCnn([blob], sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), 23.5, mix_emp_force=mix_force,
                                mix_emp_energy=mix_energy, empirical_weight=wgt, tables=ovina.Tables(), v=1000.0)
    nc.cnn_center = cen[0]
    nc.slope = 10.0
    for b in range(2):
        eo, cho = nc.eval_deriv(confs[b])
        assert abs(e[b] - eo) < 3e-4 * max(1.0, abs(eo)), (b, e[b], eo)
        assert np.abs(ch[b] - cho).max() < 3e-3 * max(np.abs(cho).max(), 1e-3), (b, ch[b], cho)
    if mix_force:
        assert np.abs(ch - ch_plain).max() > 1e-3     # the blend really changed the forces
    else:
        assert np.abs(ch - ch_plain).max() < 1e-6 and np.abs(e - e_plain / (1 + wgt)).max() < 1e-4 * np.abs(e_plain).max()


def test_monte_carlo_with_the_cnn_as_metropolis_energy(setup):
    """--cnn_scoring metrorescore / metrorefine (parallel_mc.cpp:145-155, monte_carlo.cpp:44-47): the search minimises
    on the Vina grids, the Metropolis criterion and the stored energies come from non_cache_cnn::eval.
      * one step: the first candidate is always accepted, so the chain's conformation is bit-identical to the plain
        Vina chain's (same mt19937 stream, same BFGS) while its energy is the CNN's;
      * stored energies are non_cache_cnn::eval of the stored pose (where the second BFGS did not end by reverting);
      * longer chains accept differently from the Vina chains (the reference's test_cnn.py:88-100 asserts exactly
        that the outputs differ) and are deterministic."""
    capi, sc, lig, v, olig = setup
    names = ["crossdock_default2018"]
    s = capi.Scorer(names)
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    gd = ovina.setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    v.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    box = capi.CnnBox.make(23.5, list(gd.begin), list(gd.end), slope=1e3)
    seeds = np.arange(7, 7 + 24, dtype=np.uint64)
    iters = (25 + len(lig["smt"])) // 3
    P1 = capi.McParams.default(1, iters, 10)
    n0, e0, cf0, xyz0, ev0 = v.mc_batch(seeds, list(gd.begin), list(gd.end), P1)
    n1, e1, cf1, xyz1, ev1, cnn1 = v.mc_cnn_batch(s, seeds, list(gd.begin), list(gd.end), P1, box)
    assert (n0 == 1).all() and (n1 == 1).all() and cnn1 == 2 * len(seeds)
    assert np.array_equal(cf0[:, 0], cf1[:, 0]) and np.array_equal(xyz0[:, 0], xyz1[:, 0]) and np.array_equal(ev0, ev1)
    assert not np.allclose(e0[:, 0], e1[:, 0])
    # the stored energy is the CNN igrid's energy of the stored pose (cube centred on its heavy atoms)
    co = v.coords_batch(cf1[:, 0])
    cen = np.stack([heavy_center(co[b], lig["smt"]) for b in range(len(seeds))])
    chk, _ = v.cnn_eval_batch(s, cf1[:, 0], box, cen, deriv=False)
    ok = np.abs(chk - e1[:, 0]) <= 1e-4 * np.maximum(1.0, np.abs(chk))
    assert ok.mean() >= 0.75, ok
    # longer chains
    P = capi.McParams.default(40, iters, 10)
    nA, eA, cfA, _, _, cnnA = v.mc_cnn_batch(s, seeds[:8], list(gd.begin), list(gd.end), P, box)
    nB, eB, cfB, _, _, _ = v.mc_cnn_batch(s, seeds[:8], list(gd.begin), list(gd.end), P, box)
    nV, eV, cfV, _, _ = v.mc_batch(seeds[:8], list(gd.begin), list(gd.end), P)
    assert cnnA == 2 * 40 * 8 and (nA >= 1).all()
    assert np.array_equal(eA, eB) and np.array_equal(cfA, cfB)                      # deterministic
    assert all(np.all(np.diff(eA[b, :nA[b]]) >= 0) for b in range(8))               # containers sorted by CNN energy
    assert not np.array_equal(cfA[:, 0], cfV[:, 0])                                  # the CNN accepts differently


@pytest.mark.parametrize("kind", [1, 2])
def test_cnn_metropolis_chains_under_the_accurate_line_search(setup, kind):
    """--accurate_line_search / --simple_ascent with --cnn_scoring metrorescore (minimization_params, bfgs.h:104-180,234-355;
    parallel_mc.cpp:145-155): the device chains of mi_vina_mc_cnn_batch run the handle's line search like every other
    minimiser of the handle.  One step: the first candidate is always accepted, so conformation, coordinates and
    evaluation counts are bit-identical to the plain Vina chain under the same search; they differ from the fast
    search's; longer chains are deterministic."""
    capi, sc, lig, v, olig = setup
    s = capi.Scorer(["crossdock_default2018"])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    gd = ovina.setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    v.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    box = capi.CnnBox.make(23.5, list(gd.begin), list(gd.end), slope=1e3)
    seeds = np.arange(11, 11 + 16, dtype=np.uint64)
    iters = (25 + len(lig["smt"])) // 3
    P1 = capi.McParams.default(1, iters, 10)
    lo, hi = list(gd.begin), list(gd.end)
    nf, ef, cff, _, evf, _ = v.mc_cnn_batch(s, seeds, lo, hi, P1, box)
    try:
        v.set_line_search(True, simple=(kind == 2))
        n0, e0, cf0, xyz0, ev0 = v.mc_batch(seeds, lo, hi, P1)
        n1, e1, cf1, xyz1, ev1, cnn1 = v.mc_cnn_batch(s, seeds, lo, hi, P1, box)
        assert (n0 == 1).all() and (n1 == 1).all() and cnn1 == 2 * len(seeds)
        assert np.array_equal(cf0[:, 0], cf1[:, 0]) and np.array_equal(xyz0[:, 0], xyz1[:, 0]) and np.array_equal(ev0, ev1)
        assert not np.array_equal(cf1[:, 0], cff[:, 0]) and not np.array_equal(ev1, evf)   # not the fast search's chains
        P = capi.McParams.default(12, iters, 10)
        nA, eA, cfA, _, evA, cnnA = v.mc_cnn_batch(s, seeds[:6], lo, hi, P, box)
        nB, eB, cfB, _, evB, _ = v.mc_cnn_batch(s, seeds[:6], lo, hi, P, box)
        assert cnnA == 2 * 12 * 6 and (nA >= 1).all()
        assert np.array_equal(eA, eB) and np.array_equal(cfA, cfB) and np.array_equal(evA, evB)
        assert all(np.all(np.diff(eA[b, :nA[b]]) >= 0) for b in range(6))
    finally:
        v.set_line_search(False)


@pytest.mark.parametrize("mix_force", [False, True])
def test_cnn_eval_deriv_with_a_user_grid(setup, mix_force):
    """--user_grid in non_cache_cnn::eval_deriv (non_cache_cnn.cpp:141-151): per heavy atom, curled on its own, also
    part of the empirical blend; ::eval does not see it."""
    from tests import ref_cases as RC
    capi, sc, lig, v, olig = setup
    name = "crossdock_default2018"
    s = capi.Scorer([name])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    text, _ = RC.user_grid_text(sc["center"], (14, 14, 14), 1.0, seed=8, amplitude=5.0)
    ub, ue, un, vals = capi.user_grid_parse(text)
    rng = np.random.RandomState(22)
    confs = random_confs(lig, rng, 2)
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    cen = np.stack([heavy_center(ovina.set_conf(olig, lig["conf0"])[0], lig["smt"])] * 2)
    box = capi.CnnBox.make(23.5, lo, hi, slope=10.0, mix_emp_force=mix_force, empirical_weight=0.7, v=1000.0)
    e_plain, ch_plain = v.cnn_eval_batch(s, confs, box, cen, deriv=True)
    e0_plain, _ = v.cnn_eval_batch(s, confs, box, cen, deriv=False)
    try:
        v.set_user_grid(ub, ue, un, vals, 0.5)
        e, ch = v.cnn_eval_batch(s, confs, box, cen, deriv=True)
        e0, _ = v.cnn_eval_batch(s, confs, box, cen, deriv=False)
    finally:
        v.set_user_grid(None, None, None, None)
    assert np.abs(e0 - e0_plain).max() < 1e-5 * np.abs(e0_plain).max()        # ::eval: no user grid
    assert np.abs(ch - ch_plain).max() > 1e-3
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    nc = cnn_refine.NonCacheCnn([blob], sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), 23.5, mix_emp_force=mix_force,
                                empirical_weight=0.7, tables=ovina.Tables(), v=1000.0)
    nc.cnn_center = cen[0]
    nc.slope = 10.0
    nc.user_grid = ovina.user_grid_data(ub, ue, un, vals, 0.5)
    for b in range(2):
        eo, cho = nc.eval_deriv(confs[b])
        assert abs(e[b] - eo) < 3e-4 * max(1.0, abs(eo)), (b, e[b], eo)
        assert np.abs(ch[b] - cho).max() < 3e-3 * max(np.abs(cho).max(), 1e-3), (b, ch[b], cho)


def test_monte_carlo_with_the_cnn_as_the_minimisers_igrid(setup):
    """--cnn_scoring all (parallel_mc.cpp:156-159): quasi_newton inside the search minimises non_cache_cnn, the
    Metropolis energy is non_cache_cnn::eval.  Against oracle/cnn_refine.mc_cnnall (PyTorch CPU autograd + the
    voxelizer / tree / BFGS restatements + the reference's mt19937 stream) on short chains with a two-iteration BFGS
    cap: same start, same mutation, same line searches -> the same poses where fp32 CNN gradients allow (the oracle
    and the device differ by ~1e-3 of the gradient scale, which can flip a line-search trial); plus what must hold
    regardless: determinism, sorted containers, stored energy = non_cache_cnn::eval of the stored pose."""
    capi, sc, lig, v, olig = setup
    name = "crossdock_default2018"
    s = capi.Scorer([name])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    gd = ovina.setup_grid_dims(sc["center"], sc["size"])
    lo, hi = np.array(list(gd.begin), np.float32), np.array(list(gd.end), np.float32)
    box = capi.CnnBox.make(23.5, lo, hi, slope=1e3)
    seeds = np.array([11, 12, 13], dtype=np.uint64)
    P = capi.McParams.default(2, 2, 10)
    n, e, cf, xyz, ev, cnn = v.mc_cnn_batch(s, seeds, lo, hi, P, box, level_all=True)
    n2, e2, cf2, _, ev2, _ = v.mc_cnn_batch(s, seeds, lo, hi, P, box, level_all=True)
    assert np.array_equal(e, e2) and np.array_equal(cf, cf2) and np.array_equal(ev, ev2)        # deterministic
    assert (n >= 1).all() and all(np.all(np.diff(e[b, :n[b]]) >= 0) for b in range(len(seeds)))
    assert cnn > ev.sum()                                                          # eval_deriv calls + update_energy
    co = v.coords_batch(cf[:, 0])
    cen = np.stack([heavy_center(co[b], lig["smt"]) for b in range(len(seeds))])
    chk, _ = v.cnn_eval_batch(s, cf[:, 0], box, cen, deriv=False)
    # (where the second BFGS ended by reverting, `model` -- and with it the stored energy -- sits on the last trial)
    assert (np.abs(chk - e[:, 0]) <= 1e-4 * np.maximum(1.0, np.abs(chk))).sum() >= 2
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))

    def oracle_chain(seed, steps, iters):
        nc = cnn_refine.NonCacheCnn([blob], sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), 23.5)
        nc.slope = 1e3
        return cnn_refine.mc_cnnall(nc, int(seed), steps, lo, hi, iters, num_saved=10)

    # one step, one BFGS iteration per minimisation (random start, mutation, two line searches, two update_energy
    # points, container): every chain must be the oracle's -- same number of evaluations, same pose, same energy
    P11 = capi.McParams.default(1, 1, 10)
    n1, e1, cf1, _, ev1, _ = v.mc_cnn_batch(s, seeds, lo, hi, P11, box, level_all=True)
    for b, seed in enumerate(seeds):
        out, evals = oracle_chain(seed, 1, 1)
        assert n1[b] == len(out) == 1 and ev1[b] == evals, (b, ev1[b], evals)
        assert abs(out[0][0] - e1[b, 0]) <= 5e-4 * max(1.0, abs(out[0][0])), (b, e1[b, 0], out[0][0])
        assert np.abs(out[0][1] - cf1[b, 0]).max() < 1e-3
    # two steps, two iterations: still the oracle's poses where no line-search trial flipped on a last-digit
    # difference of the CNN gradient (measured: 1-2 of these 3 seeds)
    same = 0
    for b, seed in enumerate(seeds):
        out, evals = oracle_chain(seed, 2, 2)
        if (len(out) == n[b] and abs(out[0][0] - e[b, 0]) <= 1e-3 * max(1.0, abs(out[0][0]))
                and np.abs(out[0][1] - cf[b, 0]).max() < 2e-2):
            same += 1
    assert same >= 1, same


def test_cnn_with_flexible_residues_in_eval_refinement_and_monte_carlo():
    """SURVEY 8f row 4 completed: flexible receptor residues together with the CNN inside the search.  gnina's combined
    model -- atoms [movable side chain | ligand | inflex], conf [7 + T_ligand + T_flex] -- on the reference's own
    fixture (GSK3B + test/gnina/data/flex_res_side_chain.pdbqt): DLScorer::setReceptor refreshes the side-chain rows
    from the model, getGradient returns their gradient through receptor_map (cnn_torch_scorer.cpp:208-228),
    add_minus_forces / non_cache_cnn::eval_deriv put it next to the ligand's and flex.derivative (tree.h:383-393) folds
    it into the side chain's torsions.  mi_cnn_eval_batch vs oracle/cnn_refine.py on the same model; then
    refine_structure and both CNN Monte-Carlo levels run on it and move the side chain."""
    from gnina_amd import capi
    from tests import ref_cases
    capi.init(0)
    F = np.load(os.path.join(ROOT, "tests", "golden", "real_complex.npz"))
    rigid, flex = bytes(F["rec_pdbqt"]).decode(), bytes(F["flex_pdbqt"]).decode()
    lig_text = ref_cases.long_chain_ligand(n=8, origin=(-9.0, 12.0, 3.0))
    rec_xyz, rec_smt, n_mov, n_inf = capi.read_pdbqt_receptor_flex(rigid, flex, is_text=True)
    _, _, d = capi.read_pdbqt_model(rigid, flex, lig_text, is_text=True)
    nf, le = int(d["lig_begin"]), int(d["lig_end"])
    assert nf == n_mov and le == int(d["n_movable"]) and np.array_equal(d["smt"][:nf], rec_smt[:nf])
    name = "crossdock_default2018"
    s = capi.Scorer([name])
    s.set_receptor(rec_xyz, rec_smt)
    s.set_flex(np.arange(nf, dtype=np.int32))
    v = capi.Vina()
    v.set_ligand(d)
    olig = ovina.LigandHandle(d)
    T = len(d["conf0"]) - 7
    T_flex = T - 6                                      # the 8-carbon chain has 6 torsions, the residue the rest
    assert T_flex >= 1
    rng = np.random.RandomState(5)
    confs = np.stack([d["conf0"]] * 3).astype(np.float32)
    confs[1:, :3] += rng.uniform(-0.5, 0.5, (2, 3))
    confs[1:, 7:] += rng.uniform(-0.4, 0.4, (2, T))     # ligand AND side-chain torsions move
    co = v.coords_batch(confs)
    lo = co[0, :le].min(0) - 4.0
    hi = co[0, :le].max(0) + 4.0
    smt = d["smt"]
    cen = np.stack([heavy_center(co[0, :le], smt[:le])] * 3)
    box = capi.CnnBox.make(23.5, lo, hi, slope=10.0)
    e, ch = v.cnn_eval_batch(s, confs, box, cen, deriv=True)
    e0, _ = v.cnn_eval_batch(s, confs, box, cen, deriv=False)
    assert np.abs(e - e0).max() < 1e-4 * np.abs(e).max()
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    nc = cnn_refine.NonCacheCnn([blob], rec_xyz, rec_smt, olig, (lo, hi), 23.5)
    nc.cnn_center = cen[0]
    nc.slope = 10.0
    for b in range(3):
        eo, cho = nc.eval_deriv(confs[b])
        assert abs(e[b] - eo) < 3e-4 * max(1.0, abs(eo)), (b, e[b], eo)
        assert np.abs(ch[b] - cho).max() < 3e-3 * max(np.abs(cho).max(), 1e-3), (b, ch[b], cho)
        assert np.abs(cho[6 + 6:]).max() > 1e-4        # the CNN does pull on the side chain's torsions
    # a scorer without the declaration is refused, not silently wrong
    s2 = capi.Scorer([name])
    s2.set_receptor(rec_xyz, rec_smt)
    with pytest.raises(capi.MiGninaError):
        v.cnn_eval_batch(s2, confs, box, cen, deriv=True)
    # refine_structure on the combined model: the loss goes down and the side chain's torsions take part
    start = confs[1:2].copy()
    er, cr, tries, evs = v.cnn_refine_batch(s, start, box, max_iters=8)
    e_start, _ = v.cnn_eval_batch(s, start, box, cen[:1], deriv=False)
    assert np.isfinite(er).all() and er[0] < e_start[0] and np.abs(cr[0, 13:] - start[0, 13:]).max() > 1e-4
    # CNN as the Metropolis energy (device chains) and as the minimiser's igrid too (--cnn_scoring all)
    rv_xyz, rv_smt = rec_xyz[nf + n_inf:], rec_smt[nf + n_inf:]      # the rigid part is the Vina receptor
    v.set_receptor(rv_xyz, rv_smt)
    n_pts = np.ceil((hi - lo) / 0.375).astype(np.int32)
    end = lo + n_pts * np.float32(0.375)
    types = sorted(set(int(t) for t in smt[:le] if t > 1))
    v.build_cache(list(lo), list(end), [int(k) for k in n_pts], types, 1e3)
    seeds = np.arange(3, 9, dtype=np.uint64)
    bx = capi.CnnBox.make(23.5, list(lo), list(end), slope=1e3)
    P = capi.McParams.default(4, 4, 5)
    for level_all in (False, True):
        n1, e1, cf1, xyz1, ev1, cnn1 = v.mc_cnn_batch(s, seeds, list(lo), list(end), P, bx, level_all=level_all)
        n2, e2, cf2, _, _, _ = v.mc_cnn_batch(s, seeds, list(lo), list(end), P, bx, level_all=level_all)
        assert (n1 >= 1).all() and np.isfinite(e1[:, 0]).all() and cnn1 > 0
        assert np.array_equal(n1, n2) and np.array_equal(cf1, cf2)                  # deterministic
        assert np.abs(cf1[:, 0, 13:] - d["conf0"][13:]).max() > 1e-3              # the residue's torsions were searched


def test_c5_refinement_at_96_cubed_end_to_end(setup):
    """BASELINE config C5 as configured: the Dense model at 0.25 A (96^3 grid, dense_1.3 -- the only family whose global
    pool takes another grid) with --cnn_scoring refinement, i.e. refine_structure on non_cache_cnn
    (gninasrc/main/main.cpp:131-171: quasi_newton over non_cache_cnn::eval_deriv, which is TorchModel::forward + backward +
    GridMaker::backward per evaluation, gninasrc/lib/torch_model.cpp:197-221).  mi_cnn_refine_batch runs the whole loop on
    the device; the oracle runs the same loop on the CPU (oracle/cnn_refine.py: voxelizer + CNN autograd + torsion-tree
    fold + the same bfgs<>) at the same grid."""
    capi, sc, lig, v, olig = setup
    name, RES, DIM = "dense_1_3", 0.25, 23.75
    m = capi.Model(name, resolution=RES, dimension=DIM)
    assert m.grid_points == 96
    s = capi.Scorer([m])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    rng = np.random.RandomState(21)
    confs = random_confs(lig, rng, 6, trans=0.4, rot=0.15, tors=0.3)
    lo, hi = sc["center"] - sc["size"] / 2, sc["center"] + sc["size"] / 2
    box = capi.CnnBox.make(DIM, lo, hi)
    # (i) the first two BFGS iterations against the oracle, pose by pose
    e, out, tries, evals = v.cnn_refine_batch(s, confs[:2], box, max_iters=2)
    blob = cnn_ref.Blob(os.path.join(WEIGHTS, name + ".mgw"))
    blob.resolution, blob.dimension = RES, DIM
    close = 0
    for b in range(2):
        nc = cnn_refine.NonCacheCnn([blob], sc["rec_xyz"], sc["rec_smt"], olig, (lo, hi), DIM)
        eo, co, to = cnn_refine.refine_structure(nc, confs[b], 2)
        assert to == tries[b] == 1
        print(f"C5 refine pose {b}: device {e[b]:.6f} oracle {eo:.6f}, max |d conf| {np.abs(out[b] - co).max():.2e}, evals {evals[b]} / {nc.evals}")
        assert abs(e[b] - eo) < 2e-3 * max(1.0, abs(eo))
        if abs(e[b] - eo) < 1e-3 * max(1.0, abs(eo)) and np.abs(out[b] - co).max() < 5e-3 and nc.evals == evals[b]:
            close += 1
    assert close >= 1
    # (ii) the full refinement: lower loss, inside the box, reproducible, and the refined poses re-score to the returned energies
    start, _ = v.cnn_eval_batch(s, confs, box, None, deriv=False)
    e, out, tries, evals = v.cnn_refine_batch(s, confs, box)
    assert (e <= start + 2e-5 * np.maximum(1.0, np.abs(start))).all() and (e < start - 1e-3).sum() >= 4
    assert (tries == 1).all() and (evals >= 2).all()
    after, _ = v.cnn_eval_batch(s, out, box, None, deriv=False)
    assert np.abs(after - e).max() < 1e-4 * max(1.0, float(np.abs(e).max()))
    e2, out2, _, _ = v.cnn_refine_batch(s, confs, box)
    assert np.array_equal(e, e2) and np.array_equal(out, out2)
