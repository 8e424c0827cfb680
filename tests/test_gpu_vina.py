"""GPU parity of the Vina path against the CPU oracle (oracle/vina_ref.c),
following the reference's own test pattern: random synthetic molecules, CPU vs device, abs tolerance
(test/gnina/test_gpucode.cpp:22-155, test_cache.cu:148-153, test_tree.cu:136-187 use 0.01 abs).
Bars here: tables bit-exact, cache grids 1e-5 rel, coordinates 1e-5, energy 1e-4 rel,
gradient 1e-3 rel of the gradient scale."""
import numpy as np
import pytest

from gnina_amd import synth
from oracle import vina as V
from gnina_amd import vina_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import capi as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def T():
    return V.Tables()


@pytest.fixture(scope="module")
def setup(capi, T):
    sc = vina_scene.build(0)
    lig = sc["lig"]
    gd = V.setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    grids = {t: V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], t) for t in types}
    S = V.Scene(T, gd, grids, V.LigandHandle(lig))
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    vina.set_ligand(lig)
    return vina, S, sc, gd, types, grids


def test_pair_tables_bit_exact(capi, T):
    vina = capi.Vina()
    assert vina.n == T.n == 2051
    for t1, t2 in ((2, 2), (2, 13), (7, 13), (13, 7), (0, 5), (23, 12), (27, 17)):
        f, e, d = vina.table(t1, t2)
        f0, e0, d0 = T.get(t1, t2)
        assert np.array_equal(f, f0) and np.array_equal(e, e0) and np.array_equal(d, d0)


def test_cache_grids_match_oracle(setup):
    vina, S, sc, gd, types, grids = setup
    for t in types[:4]:
        g = vina.cache_grid(t)
        assert g.shape == grids[t].shape
        assert np.array_equal(g, grids[t])       # same table entries, same (atom index) summation order: the same bits


def test_eval_deriv_matches_oracle(setup):
    vina, S, sc, gd, types, grids = setup
    rng = np.random.RandomState(11)
    lig = sc["lig"]
    confs = np.stack([synth.random_conf(rng, lig, sc["center"], spread=1.5) for _ in range(48)] + [lig["conf0"]])
    # a conformation pushed partly out of the box exercises the out-of-grid penalty branch
    out = confs[3].copy()
    out[:3] += (np.array(gd.end[:]) - np.array(gd.begin[:])) * 0.6
    confs = np.vstack([confs, out[None]])
    for v in ((1000.0, 1000.0, 1000.0), (10.0, 10.0, 10.0)):
        e, ch, co = vina.eval_batch(confs, v, deriv=True, want_coords=True)
        for b in range(len(confs)):
            e0, g0, c0, _ = S.eval_deriv(confs[b], v)
            assert np.abs(co[b] - c0).max() < 1e-4
            assert abs(e[b] - e0) <= 1e-4 * max(1.0, abs(e0)), (b, e[b], e0)
            assert np.abs(ch[b] - g0).max() <= 1e-3 * max(1.0, np.abs(g0).max()), (b, ch[b], g0)
        e2, _, _ = vina.eval_batch(confs, v, deriv=False)
        for b in range(0, len(confs), 7):
            e0 = S.eval(confs[b], v)
            assert abs(e2[b] - e0) <= 1e-4 * max(1.0, abs(e0))


def test_bfgs_step_exact_then_statistically_equivalent(setup):
    """Same start, same algorithm.  The landscape is rough (table kinks, curl caps) and BFGS amplifies
    last-ulp differences (device sinf/cosf, reduction order), so -- like the MC trajectories, SURVEY
    "Hard parts" -- parity is defined as: step-exact for short trajectories (same number of function
    evaluations, same energy), statistically equivalent for full-length ones."""
    vina, S, sc, gd, types, grids = setup
    rng = np.random.RandomState(12)
    lig = sc["lig"]
    confs = np.stack([synth.random_conf(rng, lig, sc["center"], spread=1.0) for _ in range(32)])
    for v in ((10.0, 10.0, 10.0), (1000.0, 1000.0, 1000.0)):
        e_start = vina.eval_batch(confs, v)[0]
        for iters in (1, 2, 3):
            e, cf, g, ev = vina.bfgs_batch(confs, v, max_iters=iters)
            same = 0
            for b in range(len(confs)):
                e0, c0, g0, ev0 = S.bfgs(confs[b], v, max_iters=iters)
                if ev[b] == ev0 and abs(e[b] - e0) <= 1e-3 * max(1.0, abs(e0)) and np.abs(cf[b] - c0).max() < 1e-2:
                    same += 1
            assert same >= 0.9 * len(confs), (iters, same)
        # full length (gnina: (25 + n_movable) / 3 iterations, main.cpp:454-456)
        e, cf, g, ev = vina.bfgs_batch(confs, v)
        assert (e <= e_start + 1e-5).all() and (ev >= 2).all()          # never worse than the start (bfgs.h:491-495)
        e_chk, g_chk, _ = vina.eval_batch(cf, v)                        # returned conf has the returned energy / gradient
        assert np.abs(e_chk - e).max() <= 1e-4 * max(1.0, np.abs(e).max())
        assert np.abs(g_chk - g).max() <= 1e-3 * max(1.0, np.abs(g).max())
        e_orc = np.array([S.bfgs(confs[b], v)[0] for b in range(len(confs))])
        assert abs(np.median(e) - np.median(e_orc)) <= 0.2 * abs(np.median(e_orc)) + 1.0
        assert abs(e.mean() - e_orc.mean()) <= 0.25 * abs(e_orc.mean()) + 1.0
        drop_dev, drop_orc = (e_start - e).mean(), (e_start - e_orc).mean()
        assert abs(drop_dev - drop_orc) <= 0.1 * abs(drop_orc) + 1.0
    # batch independence: one conformation alone gives the same bits as inside the batch
    e1, c1, _, _ = vina.bfgs_batch(confs[5:6], v)
    assert e1[0] == e[5] and np.array_equal(c1[0], cf[5])
    # ... and as inside a batch large enough for the throughput instantiation of the kernel (> 2,048 chains:
    # fewer look-ups in flight, H in LDS instead of registers -- same operations)
    big = np.tile(confs, (80, 1))
    eb, cb, gb, evb = vina.bfgs_batch(big, v)
    assert np.array_equal(eb[:len(confs)], e) and np.array_equal(cb[:len(confs)], cf) and np.array_equal(evb[:len(confs)], ev)
    assert np.array_equal(eb[-len(confs):], e)


def test_vina_error_paths(capi):
    v = capi.Vina()
    with pytest.raises(capi.MiGninaError):
        v.build_cache([0, 0, 0], [1, 1, 1], [2, 2, 2], [2])        # no receptor yet
    v.set_receptor(np.zeros((1, 3), dtype=np.float32), [2])
    v.build_cache([0, 0, 0], [3, 3, 3], [8, 8, 8], [2])
    v.n_tors = 0
    with pytest.raises(capi.MiGninaError):
        v.eval_batch(np.zeros((1, 7), dtype=np.float32))            # no ligand yet


def test_mc_chain_first_step_matches_oracle_and_statistics(setup, capi):
    """Row a17.  Same mt19937 stream on both sides: a 1-step chain with a short BFGS must agree
    closely (identical random start / mutation, step-exact BFGS); full chains are chaotic, so they
    are compared statistically over 48 chains.  Structural invariants of the output container
    (coords.cpp:43-56) are checked exactly."""
    vina, S, sc, gd, types, grids = setup
    c1, c2 = list(gd.begin), list(gd.end)
    seeds = np.arange(100, 148, dtype=np.uint64)
    # (a) one step, two BFGS iterations
    P = capi.McParams.default(1, 2, 8)
    n, e, cf, xyz, ev = vina.mc_batch(seeds, c1, c2, P)
    close = 0
    for b in range(len(seeds)):
        e0, cf0, xyz0, ev0 = V.mc_chain(S, c1, c2, int(seeds[b]), 1, 2, 8)
        assert n[b] == len(e0) == 1
        if abs(e[b, 0] - e0[0]) <= 1e-3 * max(1.0, abs(e0[0])) and np.abs(cf[b, 0] - cf0[0]).max() < 1e-2:
            close += 1
    assert close >= 0.8 * len(seeds), close      # two BFGS runs per step: a few starts already diverge
    # (b) longer chains: invariants + statistics
    steps, iters, saved = 120, (25 + vina.n_atoms) // 3, 20
    P = capi.McParams.default(steps, iters, saved)
    n, e, cf, xyz, ev = vina.mc_batch(seeds, c1, c2, P)
    assert (n >= 1).all() and (n <= saved).all() and (ev > steps).all()
    for b in range(0, len(seeds), 5):
        eb = e[b, :n[b]]
        assert np.all(np.diff(eb) >= 0)                                   # container sorted by energy
        # stored e = cache::eval on what `model` held after the second minimisation: the stored conformation itself,
        # unless that BFGS ended by reverting to its start (then the last line-search trial, monte_carlo.cpp:44-47)
        chk = vina.eval_batch(cf[b, :n[b]], grid_only=True, deriv=False)[0]
        ok = np.abs(chk - eb) <= 1e-4 * np.maximum(1.0, np.abs(eb))
        assert ok.mean() >= 0.8, (b, ok)
        co = vina.eval_batch(cf[b, :n[b]], want_coords=True)[2]
        heavy = np.nonzero(sc["lig"]["smt"] > 1)[0]
        assert np.abs(co[:, heavy] - xyz[b, :n[b]]).max() < 1e-3           # stored coords = heavy atoms of conf
    best_dev = e[:, 0]
    orc = [V.mc_chain(S, c1, c2, int(s), steps, iters, saved) for s in seeds]
    best_orc = np.array([o[0][0] for o in orc])
    ev_orc = np.array([o[3] for o in orc])
    sem = np.sqrt(best_dev.var() / len(seeds) + best_orc.var() / len(seeds))
    assert abs(best_dev.mean() - best_orc.mean()) <= 4 * sem + 0.05 * abs(best_orc.mean()), (best_dev.mean(), best_orc.mean(), sem)
    assert abs(ev.mean() - ev_orc.mean()) <= 0.15 * ev_orc.mean()          # same amount of optimisation work
    # (b') strict-order mode: the device chains ARE the restatement's (= the reference's, tests/test_ref_vina.py) chains --
    # every saved pose's energy, conformation, coordinates and the evaluation count, bit for bit.  This also covers the
    # container entries the 80 % invariant above leaves unchecked (stored energy taken on a reverted BFGS's last trial).
    vina.set_strict_order(True)
    try:
        ns, es, cfs, xyzs, evs = vina.mc_batch(seeds, c1, c2, P)
    finally:
        vina.set_strict_order(False)
    for b in range(len(seeds)):
        e0, cf0, xyz0, ev0 = orc[b]
        k = len(e0)
        assert ns[b] == k and evs[b] == ev0, (b, ns[b], k, evs[b], ev0)
        assert np.array_equal(es[b, :k], e0) and np.array_equal(cfs[b, :k], cf0) and np.array_equal(xyzs[b, :k], xyz0), b
    # (c) determinism: same seeds -> same bits; different seeds -> different chains
    n2, e2, _, _, _ = vina.mc_batch(seeds[:4], c1, c2, P)
    assert np.array_equal(e2[:, 0], e[:4, 0])
    n3, e3, _, _, _ = vina.mc_batch(seeds[:4] + np.uint64(1000), c1, c2, P)
    assert not np.array_equal(e3[:, 0], e[:4, 0])


def test_noncache_exact_and_refine_match_oracle(setup):
    """Row a18 pieces: the direct (non_cache) receptor term, precalculate_exact, refine_structure's
    slope ladder and the final reported energies (main.cpp:131-171,339-344)."""
    vina, S, sc, gd, types, grids = setup
    rng = np.random.RandomState(21)
    lig = sc["lig"]
    confs = np.stack([synth.random_conf(rng, lig, sc["center"], spread=1.0) for _ in range(10)] + [lig["conf0"]])
    out = confs[2].copy()
    out[:3] += (np.array(gd.end[:]) - np.array(gd.begin[:])) * 0.55
    confs = np.vstack([confs, out[None]])
    v = (1000.0, 1000.0, 1000.0)
    for exact in (False, True):
        e, ch, _ = vina.eval_batch(confs, v, deriv=True, direct=True, exact=exact)
        e0s = []
        for b in range(len(confs)):
            e0, g0, inter, intra = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], confs[b], v, deriv=True, exact=exact)
            assert abs(e[b] - e0) <= 2e-4 * max(1.0, abs(e0)), (exact, b, e[b], e0)
            # the exact derivative is a difference over 1e-5 A in fp32: noisy by construction
            tol = (5e-2 if exact else 1e-3) * max(1.0, np.abs(g0).max())
            assert np.abs(ch[b] - g0).max() <= tol, (exact, b)
        e_only = vina.eval_batch(confs, v, deriv=False, direct=True, exact=exact)[0]
        for b in range(0, len(confs), 3):
            e0 = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], confs[b], v, deriv=False, exact=exact)[0]
            assert abs(e_only[b] - e0) <= 2e-4 * max(1.0, abs(e0))
    # final energies (do_search's docking branch, main.cpp:231,339-344): receptor term = non_cache on the LINEAR
    # tables, pair terms exact, then num_tors_div -- the oracle recipe is pinned to the reference in test_ref_vina.py
    num_tors = 6.0
    ef, intra = vina.final_energies(confs, num_tors, v)
    for b in range(len(confs)):
        _, _, inter_l, _ = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], confs[b], v, deriv=False, exact=False)
        _, _, _, intra0 = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], confs[b], v, deriv=False, exact=True)
        assert abs(intra[b] - intra0) <= 2e-4 * max(1.0, abs(intra0))
        ref = V.conf_independent(np.float32(np.float32(inter_l) + np.float32(intra0)) - np.float32(intra0), num_tors)
        assert abs(ef[b] - ref) <= 5e-4 * max(1.0, abs(ref))
    # refine_structure: everything ends inside the box (or reports max_fl), energies never increase
    e_start = vina.eval_batch(confs, v, deriv=False, direct=True)[0]
    er, cr, tries = vina.refine_batch(confs, v)
    assert ((tries >= 1) & (tries <= 5)).all()
    co = vina.eval_batch(cr, v, want_coords=True)[2]
    heavy = lig["smt"] > 1
    for b in range(len(confs)):
        inside = ((co[b][heavy] >= np.array(gd.begin[:]) - 1e-3) & (co[b][heavy] <= np.array(gd.end[:]) + 1e-3)).all()
        assert inside == (er[b] < 1e30)
    assert tries[-1] >= 1 and er[-1] < 1e30            # the out-of-box pose was pulled back in
    e_orc = np.array([V.refine(S, sc["rec_xyz"], sc["rec_smt"], confs[b], v)[0] for b in range(len(confs))])
    fin = (er < 1e30) & (e_orc < 1e30)
    assert abs(np.median(er[fin]) - np.median(e_orc[fin])) <= 0.25 * abs(np.median(e_orc[fin])) + 1.0


def test_pdbqt_ligand_through_the_engine(capi, T):
    """A ligand read by the native PDBQT reader (torsion tree, types, pairs) drives the Vina kernels and the CNN
    exactly like the synthetic generator's ligands: device == oracle on the same description."""
    from tests.test_pdbqt_cpu import chain_pdbqt
    lig = capi.read_pdbqt_ligand(chain_pdbqt(), is_text=True)
    sc = vina_scene.build(seed=2)
    shift = sc["center"] - lig["coords0"].mean(0)
    conf = lig["conf0"].copy()
    conf[:3] += shift
    gd = V.setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    v = capi.Vina()
    v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    v.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    v.set_ligand(lig)
    rng = np.random.RandomState(4)
    confs = np.stack([conf] * 4).astype(np.float32)
    confs[1:, 7:] += rng.uniform(-2, 2, (3, lig["n_tors"])).astype(np.float32)
    confs[1:, :3] += rng.uniform(-1, 1, (3, 3)).astype(np.float32)
    e, ch, coords = v.eval_batch(confs, (1000.0, 1000.0, 1000.0), deriv=True, want_coords=True)
    grids = {t: V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], t) for t in types}
    S = V.Scene(T, gd, grids, V.LigandHandle(lig))
    for b in range(4):
        eo, cho, co, _ = S.eval_deriv(confs[b])
        assert np.abs(coords[b] - co).max() < 1e-4
        assert abs(e[b] - eo) < 1e-4 * max(1.0, abs(eo))
        assert np.abs(ch[b] - cho).max() < 2e-3 * max(np.abs(cho).max(), 1e-2)
    assert np.abs(coords[0] - (lig["coords0"] + shift)).max() < 1e-4      # conf0 reproduces the file's pose
    s = capi.Scorer(["crossdock_default2018"])
    s.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    out = s.score_batch(coords, lig["smt"])
    assert np.isfinite(out["pose"]).all() and np.isfinite(out["affinity"]).all()


def test_mc_output_is_independent_of_the_team_size(setup, capi, monkeypatch):
    """The Monte-Carlo kernel runs W wavefronts per chain that evaluate the line-search trials of one search
    at once and keep the first accepted trial in trial order (vina.hip, bfgs_wave): energies, conformations
    and the reference-equivalent evaluation counts must be the same bits for W = 1, 2, 4."""
    vina, S, sc, gd, types, grids = setup
    c1, c2 = list(gd.begin), list(gd.end)
    seeds = np.arange(500, 516, dtype=np.uint64)
    P = capi.McParams.default(60, (25 + vina.n_atoms) // 3, 10)
    ref = None
    for w in ("1", "2", "4"):
        capi.set_option("MI_VINA_MC_WAVES", w)
        out = vina.mc_batch(seeds, c1, c2, P)
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), w
    capi.set_option("MI_VINA_MC_WAVES", None)
    assert (ref[4] > 60).all()


@pytest.mark.parametrize("n_atoms,n_tors", [(60, 26), (140, 2)])
def test_large_ligand_paths_match_oracle(capi, T, n_atoms, n_tors):
    """Ligands past the register fast paths.  (60 atoms, 20 torsions): 6 + T = 26 > 24 variables, so the
    inverse-Hessian estimate lives in LDS instead of registers, and ~1,500 interacting pairs = several groups of
    pair look-ups per lane and a workspace above the default 64 KB of dynamic LDS.  (140 atoms, 2 torsions): the
    grid look-ups of the atoms beyond the first 128 take the fused path, three atoms per lane."""
    sc = vina_scene.build(3, n_rec=1500, n_atoms=n_atoms, n_tors=n_tors, pad=3.0)
    lig = sc["lig"]
    gd = V.setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    grids = {t: V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], t) for t in types}
    S = V.Scene(T, gd, grids, V.LigandHandle(lig))
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    vina.set_ligand(lig)
    assert len(lig["pairs"]) > 64 * 6 and (vina.n_tors >= 19 or n_atoms > 128)
    rng = np.random.RandomState(5)
    confs = np.stack([synth.random_conf(rng, lig, sc["center"], spread=0.5) for _ in range(6)] + [lig["conf0"]])
    v = (10.0, 10.0, 10.0)
    e, ch, co = vina.eval_batch(confs, v, deriv=True, want_coords=True)
    for b in range(len(confs)):
        e0, g0, c0, _ = S.eval_deriv(confs[b], v)
        assert np.abs(co[b] - c0).max() < 2e-4
        assert abs(e[b] - e0) <= 2e-4 * max(1.0, abs(e0)), (b, e[b], e0)
        assert np.abs(ch[b] - g0).max() <= 2e-3 * max(1.0, np.abs(g0).max()), b
    e1, cf1, g1, ev1 = vina.bfgs_batch(confs, v, max_iters=2)
    same = 0
    for b in range(len(confs)):
        e0, c0, g0, ev0 = S.bfgs(confs[b], v, max_iters=2)
        if ev1[b] == ev0 and abs(e1[b] - e0) <= 1e-3 * max(1.0, abs(e0)) and np.abs(cf1[b] - c0).max() < 1e-2:
            same += 1
    assert same >= len(confs) - 2, same
    n, em, cfm, xyz, ev = vina.mc_batch(np.arange(3, dtype=np.uint64), list(gd.begin), list(gd.end),
                                        capi.McParams.default(3, 4, 4))
    assert (n >= 1).all() and np.isfinite(em[:, 0]).all()


def test_screen_launch_matches_per_ligand_chains(capi, T):
    """mi_vina_mc_screen docks the chains of several different ligands in ONE launch; every chain must give the
    same bits as mi_vina_mc_batch of its ligand with the same seed (per-ligand step / iteration counts,
    containers at the common strides)."""
    sc = vina_scene.build(0)
    rng = np.random.RandomState(3)
    ligs = []
    for na, nt in ((18, 2), (32, 6), (26, 9)):
        lig = synth.make_ligand_tree(rng, na, nt)
        shift = -lig["coords0"].mean(0)
        lig["coords0"] = (lig["coords0"] + shift).astype(np.float32)
        lig["conf0"][:3] += shift
        ligs.append(lig)
    gd = V.setup_grid_dims(np.zeros(3, np.float32), np.full(3, 18.0, np.float32))
    types = sorted({int(t) for lig in ligs for t in lig["smt"] if t > 1})
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    c1, c2 = list(gd.begin), list(gd.end)
    params = [capi.McParams.default(20 + 7 * l, (25 + len(lig["smt"])) // 3, 6) for l, lig in enumerate(ligs)]
    chain_lig = np.array([0, 1, 2, 2, 1, 0, 1], dtype=np.int32)
    seeds = np.arange(900, 900 + len(chain_lig), dtype=np.uint64)
    vina.set_screen(ligs)
    n, e, cf, xyz, ev = vina.mc_screen(chain_lig, seeds, c1, c2, params)
    assert (n >= 1).all()
    for l, lig in enumerate(ligs):
        vina.set_ligand(lig)
        idx = np.nonzero(chain_lig == l)[0]
        n1, e1, cf1, xyz1, ev1 = vina.mc_batch(seeds[idx], c1, c2, params[l])
        nc, nh3 = 7 + lig["n_tors"], 3 * int((lig["smt"] > 1).sum())
        assert np.array_equal(n[idx], n1) and np.array_equal(ev[idx], ev1)
        for k, b in enumerate(idx):                       # rows beyond n are unspecified
            m = n1[k]
            assert np.array_equal(e[b, :m], e1[k, :m])
            assert np.array_equal(cf[b, :m, :nc], cf1[k, :m])
            assert np.array_equal(xyz[b, :m, :nh3], xyz1[k, :m].reshape(m, nh3))
    with pytest.raises(capi.MiGninaError):
        vina.mc_screen(np.array([3], np.int32), seeds[:1], c1, c2, params)      # ligand index out of range
    # the per-ligand tail (refine_structure, coordinates, final energies) of all ligands' poses in one launch each
    item = np.array([2, 0, 1, 1, 2], dtype=np.int32)
    rows = np.zeros((len(item), vina.screen_conf), np.float32)
    for k, l in enumerate(item):
        rows[k, :7 + ligs[l]["n_tors"]] = synth.random_conf(np.random.RandomState(40 + k), ligs[l], np.zeros(3), 1.0)
    iters = [(25 + len(lig["smt"])) // 3 for lig in ligs]
    v3 = (1000.0, 1000.0, 1000.0)
    es, chs, cos = vina.eval_screen(item, rows, v3, deriv=True, want_coords=True)
    er, rcf, tr = vina.refine_screen(item, rows, iters)
    ef, intra = vina.final_energies_screen(item, rcf, [float(lig["n_tors"]) for lig in ligs])
    for l, lig in enumerate(ligs):
        vina.set_ligand(lig)
        idx = np.nonzero(item == l)[0]
        nc, na = 7 + lig["n_tors"], len(lig["smt"])
        e1, ch1, co1 = vina.eval_batch(rows[idx][:, :nc], v3, deriv=True, want_coords=True)
        assert np.array_equal(es[idx], e1) and np.array_equal(chs[idx][:, :nc - 1], ch1)
        assert np.array_equal(cos[idx][:, :na], co1)
        e2, cf2, t2 = vina.refine_batch(rows[idx][:, :nc], max_iters=iters[l])
        assert np.array_equal(er[idx], e2) and np.array_equal(rcf[idx][:, :nc], cf2) and np.array_equal(tr[idx], t2)
        e3, i3 = vina.final_energies(cf2, float(lig["n_tors"]))
        assert np.array_equal(ef[idx], e3) and np.array_equal(intra[idx], i3)


def test_vina_pool_splits_chains_by_chain_id_and_returns_the_single_handle_bits(setup, capi, allow_duplicate_devices):
    """mi_vina_pool (include/mi_gnina.h): parallel_mc's fan-out of chains (parallel_mc.cpp:183-214) over devices.  On a
    one-GPU box MI_POOL_ALLOW_DUPLICATE_DEVICES lets three workers share device 0 -- every shard offset of the chain
    arrays is exercised -- and a chain depends on its seed and the handle's state only, so the pool must return the bits
    of the single handle, whatever the number of workers (19 chains over 3 workers: shards of 6, 6 and 7)."""
    vina, S, sc, gd, types, grids = setup
    c1, c2 = list(gd.begin), list(gd.end)
    seeds = np.arange(900, 919, dtype=np.uint64)
    P = capi.McParams.default(40, (25 + vina.n_atoms) // 3, 10)
    ref = vina.mc_batch(seeds, c1, c2, P)
    ndev = capi.lib().mi_gnina_device_count()
    devices = list(range(ndev)) if ndev >= 2 else [0, 0, 0]
    pool = capi.VinaPool(devices)

    def configure(v, rank):  # the same set-up on every device
        v.set_receptor(sc["rec_xyz"], sc["rec_smt"])
        v.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
        v.set_ligand(sc["lig"])

    pool.configure(configure)
    out = pool.mc_batch(seeds, c1, c2, P)
    assert np.array_equal(ref[0], out[0]) and np.array_equal(ref[4], out[4])          # saved minima, evaluation counts
    for b in range(len(seeds)):                                                       # (rows past out_n are not written)
        k = ref[0][b]
        for a in (1, 2, 3):
            assert np.array_equal(ref[a][b, :k], out[a][b, :k]), (b, a)
    assert pool.info()["ranks"] == len(devices)
    # an error inside the callback of one rank is reported, and the pool survives it
    with pytest.raises(Exception):
        pool.configure(lambda v, rank: v.set_ligand({"bogus": 1}) if rank == len(devices) - 1 else None)
    out2 = pool.mc_batch(seeds[:5], c1, c2, P)
    assert np.array_equal(out2[0], ref[0][:5]) and np.array_equal(out2[1][:, 0], ref[1][:5, 0])
