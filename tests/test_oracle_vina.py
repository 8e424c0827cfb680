"""Checks of the Vina oracle (oracle/vina_ref.c).  The reference stores no numbers for this path
("parity unpinned"), so the restatement is held to the construction rules of the cited
source lines and to analytic identities (finite-difference gradients, invariances)."""
import numpy as np
import pytest

from gnina_amd import synth
from oracle import vina as V
from gnina_amd import vina_scene

C_H, N_D, O_A, HYD = 2, 7, 13, 0


@pytest.fixture(scope="module")
def T():
    return V.Tables()


def test_terms_known_values():
    w = [-0.035579, -0.005156, 0.840245, -0.035069, -0.587439]  # main.cpp:1324-1328
    # C-C at the optimal distance 3.8 A: gauss1 = 1, gauss2 = exp(-(3/2)^2), no repulsion, hydrophobic = 1
    e = V.pair_energy(C_H, C_H, 3.8)
    assert abs(e - (w[0] + w[1] * np.exp(-2.25) + w[3])) < 1e-7
    # donor N - acceptor O at 2.8 A (d = -0.7): full h-bond, repulsion 0.49
    e = V.pair_energy(N_D, O_A, 2.8)
    d = 2.8 - 3.5
    ref = w[0] * np.exp(-(d / 0.5) ** 2) + w[1] * np.exp(-((d - 3) / 2) ** 2) + w[2] * d * d + w[4]
    assert abs(e - ref) < 1e-6
    assert V.pair_energy(C_H, O_A, 20.0) == pytest.approx(0.0, abs=1e-12)   # everything has decayed
    assert V.pair_energy(C_H, N_D, 4.0) == V.pair_energy(N_D, C_H, 4.0)


def test_table_construction_rules(T):
    assert T.n == int(32 * 64) + 3 == 2051                           # precalculate.h:189
    fast, se, sd = T.get(C_H, O_A)
    rs = np.sqrt(np.arange(T.n + 2, dtype=np.float32) / np.float32(32))
    for i in (0, 1, 100, 1000, 2050):
        assert se[i] == np.float32(V.pair_energy(C_H, O_A, float(rs[i])))
    assert sd[0] == 0 and sd[-1] == 0                                # precalculate.h:141-142
    i = 700
    assert sd[i] == np.float32((se[i + 1] - se[i - 1]) / ((rs[i + 1] - rs[i - 1]) * rs[i]))
    assert fast[i] == np.float32((se[i + 1] + se[i]) / np.float32(2))
    assert fast[-1] == np.float32(se[-1] / np.float32(2))            # f2 = 0 past the end
    f2, se2, sd2 = T.get(O_A, C_H)                                   # symmetric storage
    assert np.array_equal(se, se2) and np.array_equal(sd, sd2)
    # lookup = linear interpolation in r^2 (precalculate.h:97-133)
    r2 = 13.37
    e, dor = T.eval_deriv(C_H, O_A, r2)
    x = np.float32(32) * np.float32(r2)
    i1 = int(x)
    rem = np.float32(x - np.float32(i1))
    assert e == np.float32(se[i1] + rem * (se[i1 + 1] - se[i1]))
    assert dor == np.float32(sd[i1] + rem * (sd[i1 + 1] - sd[i1]))
    assert T.eval_fast(C_H, O_A, r2) == fast[i1]
    # dor really is (dE/dr)/r: compare with a finite difference of the analytic term
    r = np.sqrt(r2)
    fd = (V.pair_energy(C_H, O_A, r + 1e-3) - V.pair_energy(C_H, O_A, r - 1e-3)) / 2e-3 / r
    assert abs(dor - fd) < 5e-4


def test_grid_dims_and_populate(T):
    gd = V.setup_grid_dims([1.0, -2.0, 0.5], [10.0, 11.0, 12.2])
    assert list(gd.n) == [27, 30, 33]                                # ceil(size / 0.375)
    assert gd.end[0] - gd.begin[0] == pytest.approx(27 * 0.375)
    rng = np.random.RandomState(1)
    rec = rng.uniform(-8, 8, (60, 3)).astype(np.float32)
    smt = rng.choice([2, 6, 7, 13, 1], 60).astype(np.int32)
    g = V.cache_populate(T, gd, rec, smt, O_A)
    assert g.shape == (34, 31, 28)
    # brute force at one grid point (x fastest: data[z][y][x]); receptor hydrogens are not interaction partners
    # (szv_grid drops them, szv_grid.h:69-73) and this box is not aligned with szv_grid's 3 A cells
    x, y, z = 5, 17, 20
    p = np.array([gd.begin[0] + x * 0.375, gd.begin[1] + y * 0.375, gd.begin[2] + z * 0.375])
    acc = np.float32(0)
    for a in range(60):
        r2 = float(((rec[a] - p.astype(np.float32)) ** 2).sum(dtype=np.float32))
        if r2 <= 64 and smt[a] > 1:
            acc = np.float32(acc + np.float32(T.eval_fast(int(smt[a]), O_A, r2)))
    assert abs(g[z, y, x] - acc) < 1e-5


def test_grid_evaluate_trilinear_gradient_and_penalty(T):
    gd = V.setup_grid_dims([0, 0, 0], [6, 6, 6])
    rng = np.random.RandomState(2)
    data = rng.normal(0, 0.3, gd.shape).astype(np.float32) - 0.5     # mostly negative: no curl
    # on a grid point the value is the stored value
    loc = np.array([gd.begin[0] + 3 * 0.375, gd.begin[1] + 4 * 0.375, gd.begin[2] + 5 * 0.375], dtype=np.float32)
    e, d = V.grid_evaluate(gd, data, loc, 1e3, 1000.0)
    if data[5, 4, 3] <= 0:
        assert abs(e - data[5, 4, 3]) < 1e-5
    # analytic gradient vs finite differences inside a cell
    loc = np.array([0.33, -1.21, 0.77], dtype=np.float32)
    e, d = V.grid_evaluate(gd, data, loc, 1e3, 1e30)
    for k in range(3):
        lp, lm = loc.copy(), loc.copy()
        lp[k] += 1e-3
        lm[k] -= 1e-3
        fd = (V.grid_evaluate(gd, data, lp, 1e3, 1e30)[0] - V.grid_evaluate(gd, data, lm, 1e3, 1e30)[0]) / 2e-3
        assert abs(fd - d[k]) < 2e-2
    # out of the box: linear penalty slope * distance, gradient +-slope (grid.cpp:103-127,176-181)
    out = np.array([gd.end[0] + 0.5, 0, 0], dtype=np.float32)
    edge = np.array([gd.end[0], 0, 0], dtype=np.float32)
    e_out, d_out = V.grid_evaluate(gd, data, out, 10.0, 1e30)
    e_in, _ = V.grid_evaluate(gd, data, edge - 1e-4, 10.0, 1e30)
    assert abs((e_out - e_in) - 10.0 * 0.5) < 5e-2 and d_out[0] == pytest.approx(10.0)
    # curl: positive energies are squashed to v * e / (v + e)  (curl.h:29-42)
    pos = np.full(gd.shape, 3.0, dtype=np.float32)
    e, _ = V.grid_evaluate(gd, pos, loc, 1e3, 10.0)
    assert abs(e - 3.0 * 10 / 13) < 1e-5


@pytest.fixture(scope="module")
def scene(T):
    sc = vina_scene.build(0)
    gd = V.setup_grid_dims(sc["center"], sc["size"])
    lig = sc["lig"]
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    grids = {t: V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], t) for t in types}
    return V.Scene(T, gd, grids, V.LigandHandle(lig)), sc


def test_set_conf_rigid_motion_and_torsion(scene):
    S, sc = scene
    lig = sc["lig"]
    c0, _, _ = V.set_conf(S.lig, lig["conf0"])
    assert np.abs(c0 - lig["coords0"]).max() < 1e-5
    conf = synth.random_conf(np.random.RandomState(3), lig, sc["center"])
    c1, origin, axis = V.set_conf(S.lig, conf)
    # bonded geometry inside every rigid node is preserved
    for k in range(len(lig["parent"])):
        a, b = lig["abeg"][k], lig["aend"][k]
        if b - a >= 2:
            d0 = np.linalg.norm(c0[a:b, None] - c0[None, a:b], axis=-1)
            d1 = np.linalg.norm(c1[a:b, None] - c1[None, a:b], axis=-1)
            assert np.abs(d0 - d1).max() < 1e-4
    assert np.abs(np.linalg.norm(axis[1:], axis=1) - 1).max() < 1e-5


def test_eval_deriv_matches_finite_differences(scene):
    S, sc = scene
    lig = sc["lig"]
    rng = np.random.RandomState(4)
    v = (1000.0, 1000.0, 1000.0)
    n = 6 + lig["n_tors"]
    for trial in range(3):
        conf = synth.random_conf(rng, lig, sc["center"], spread=1.0)
        e, g, coords, forces = S.eval_deriv(conf, v)
        assert np.isfinite(e) and np.isfinite(g).all()
        fd = np.zeros(n)
        for i in range(n):
            p = np.zeros(n, dtype=np.float32)
            p[i] = 1
            h = 5e-4   # the landscape is rough (table kinks, curl): FD converges only for small steps
            ep = S.eval_deriv(V.conf_increment(conf, p, h, lig["n_tors"]), v)[0]
            em = S.eval_deriv(V.conf_increment(conf, p, -h, lig["n_tors"]), v)[0]
            fd[i] = (ep - em) / (2 * h)
        # Vina's "derivative" is the interpolated central-difference table (dor), not the exact
        # gradient of the piecewise-linear energy, and the energy has kinks (curl, box penalty):
        # agreement is to a few percent of the gradient scale, like the reference's own 0.01-abs tests.
        assert np.abs(fd - g).max() < 0.03 * np.abs(g).max() + 0.1, (trial, fd, g)
        assert np.corrcoef(fd, g)[0, 1] > 0.999


def test_energy_only_path_is_close_to_deriv_path(scene):
    """model::eval uses the midpoint table (eval_fast) for the pairs, eval_deriv interpolates:
    the two agree to table resolution."""
    S, sc = scene
    conf = synth.random_conf(np.random.RandomState(5), sc["lig"], sc["center"], spread=0.5)
    assert abs(S.eval(conf) - S.eval_deriv(conf)[0]) < 0.05 * max(1.0, abs(S.eval(conf)))


def test_bfgs_descends_and_is_idempotent_at_a_minimum(scene):
    S, sc = scene
    rng = np.random.RandomState(6)
    for v in ((10.0, 10.0, 10.0), (1000.0, 1000.0, 1000.0)):      # hunt cap / authentic v (main.cpp:460)
        conf = synth.random_conf(rng, sc["lig"], sc["center"], spread=1.0)
        e0 = S.eval_deriv(conf, v)[0]
        e1, c1, g1, evals = S.bfgs(conf, v)
        assert e1 <= e0 and evals >= 2
        assert abs(S.eval_deriv(c1, v)[0] - e1) < 1e-4 * max(1.0, abs(e1))
        e2, c2, _, _ = S.bfgs(c1, v, max_iters=3)
        assert e2 <= e1 + 1e-6
    # quaternion stays normalised through many increments
    assert abs(np.linalg.norm(c1[3:7]) - 1) < 1e-3


def test_noncache_agrees_with_cache_inside_the_box(scene):
    """The cache grid is a trilinear interpolation of the same pair sum non_cache evaluates directly:
    inside the box the two receptor terms agree to grid resolution (this is why gnina can dock on the
    grids and refine / report on non_cache)."""
    S, sc = scene
    rng = np.random.RandomState(8)
    v = (1000.0, 1000.0, 1000.0)
    diffs = []
    for _ in range(6):
        conf = synth.random_conf(rng, sc["lig"], sc["center"], spread=0.5)
        e_cache = S.eval_deriv(conf, v)[0]
        e_nc, g_nc, inter, intra = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], conf, v)
        assert abs(inter + intra - e_nc) < 1e-3 * max(1.0, abs(e_nc))
        diffs.append(abs(e_cache - e_nc) / max(1.0, abs(e_nc)))
    assert np.median(diffs) < 0.15


def test_exact_precalculate_is_close_to_tables_and_gradient_consistent(scene):
    S, sc = scene
    conf = synth.random_conf(np.random.RandomState(9), sc["lig"], sc["center"], spread=0.5)
    v = (1000.0, 1000.0, 1000.0)
    e_t, g_t, _, _ = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], conf, v, exact=False)
    e_x, g_x, _, _ = V.noncache_eval(S, sc["rec_xyz"], sc["rec_smt"], conf, v, exact=True)
    assert abs(e_t - e_x) < 0.02 * max(1.0, abs(e_x))
    assert np.abs(g_t - g_x).max() < 0.1 * max(1.0, np.abs(g_x).max())
    # num_tors_div: e / (1 + 0.05846 * num_tors)  
    assert V.conf_independent(-10.0, 6.0) == pytest.approx(-10.0 / (1 + 0.05846 * 6), rel=1e-5)


def test_refine_structure_pulls_a_pose_back_into_the_box(scene):
    S, sc = scene
    gd = S.gd
    conf = sc["lig"]["conf0"].copy()
    conf[:3] += (np.array(gd.end[:]) - np.array(gd.begin[:])) * 0.55      # mostly outside the box
    e, c1, tries = V.refine(S, sc["rec_xyz"], sc["rec_smt"], conf)
    assert 1 <= tries <= 5
    coords, _, _ = V.set_conf(S.lig, c1)
    heavy = sc["lig"]["smt"] > 1
    inside = ((coords[heavy] >= np.array(gd.begin[:]) - 1e-3) & (coords[heavy] <= np.array(gd.end[:]) + 1e-3)).all()
    assert inside == (e < 1e30)
