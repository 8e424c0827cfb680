"""Pins the Vina/smina path to THE REFERENCE ITSELF: oracle/_ref/libgnina_ref.so is gnina's own parse_pdbqt.cpp,
model.cpp/.cu, tree.h, conf.h, everything.h, weighted_terms.cpp, precalculate.h, cache.cpp, grid.cpp, szv_grid.h,
non_cache.cpp, bfgs.h, quasi_newton.cpp, monte_carlo.cpp, mutate.cpp ... compiled unmodified from /root/reference
behind stand-in headers (oracle/ref_shims, recipe oracle/Makefile.ref).  Checked against it here, on the same PDBQT
bytes:
  * gnina_amd/host/pdbqt.cpp (atom order, types, pairs, trees, num_tors, error behaviour)        -- bit-exact
  * oracle/vina_ref.c, the restatement the GPU parity tests use (rows a11-a18 of SURVEY 8)       -- bit-exact:
    tables, cache grids (incl. szv_grid's first-point candidate bricks), eval / eval_deriv on cache and non_cache,
    exact pair terms, conf increment, whole BFGS runs, whole Monte-Carlo chains (same mt19937 stream).
CPU only; needs /root/reference (skipped elsewhere).  tests/golden/vina_goldens.npz freezes reference outputs of the
same cases for the GPU tests (tests/golden/make_vina_goldens.py)."""
import os

import numpy as np
import pytest

from oracle import ref, vina as V
from tests import ref_cases as RC

pytestmark = pytest.mark.skipif(not os.path.isdir(RC.REF_DATA) or not ref.available(),
                                reason="needs /root/reference (oracle/_ref is built from it)")
V3 = (1000.0, 1000.0, 1000.0)
HUNT = (10.0, 10.0, 10.0)


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import build, capi as c
    build.build()
    return c


@pytest.fixture(scope="module")
def rigid_text():
    return open(RC.GSK3B).read()


class Case:
    """reference scene + our reader's ligand + the oracle scene on the same box"""

    def __init__(self, capi, rigid_text, lig_text, center=None, size=None):
        self.ref = ref.Scene(rigid_text, lig_text)
        self.lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
        self.rec_xyz, self.rec_smt = self.ref.grid_atoms()
        if center is None:
            center, size = RC.box_of(self.lig["coords0"])
        self.begin, self.end, self.n = self.ref.build_grids(center, size)
        self.gd = V.setup_grid_dims(center, size)
        self.T = V.Tables()
        self.h = V.LigandHandle(self.lig)
        types = sorted(set(int(t) for t in self.lig["smt"] if t > 1))
        self.grids = {t: V.cache_populate(self.T, self.gd, self.rec_xyz, self.rec_smt, t) for t in types}
        self.ora = V.Scene(self.T, self.gd, self.grids, self.h)
        self.max_iters = (25 + self.ref.n_movable) // 3


@pytest.fixture(scope="module")
def adduct(capi, rigid_text):
    return Case(capi, rigid_text, RC.cys_adduct_ligand())


@pytest.fixture(scope="module")
def chain(capi, rigid_text):
    return Case(capi, rigid_text, RC.long_chain_ligand())


def test_receptor_types_and_coordinates_are_the_references(capi, rigid_text):
    s = ref.Scene(rigid_text)
    gx, gs = s.grid_atoms()
    rx, rs = capi.read_pdbqt_receptor(RC.GSK3B)
    assert len(gs) == 3460 and np.array_equal(gs, rs) and np.array_equal(gx, rx)
    assert (gs <= 1).sum() > 300          # the file has polar hydrogens: they matter below


@pytest.mark.parametrize("which", ["adduct", "chain", "hand"])
def test_ligand_reader_matches_the_reference(capi, rigid_text, which, request):
    if which == "hand":
        from tests.test_pdbqt_cpu import chain_pdbqt
        c = Case(capi, rigid_text, chain_pdbqt())
    else:
        c = request.getfixturevalue(which)
    s, lig = c.ref, c.lig
    xyz, smt, _ = s.atoms()
    assert s.n_atoms == len(lig["smt"]) and s.n_lig_tors == lig["n_tors"]
    assert np.array_equal(smt, lig["smt"]) and np.array_equal(xyz, lig["coords0"])     # atom order, adjust_smina_type
    pairs, t12 = s.pairs()
    assert np.array_equal(pairs, lig["pairs"])                                          # initialize_pairs, in order
    assert np.array_equal(s.initial_conf(), lig["conf0"])
    # conf_independent_inputs::num_tors through the num_tors_div term
    assert s.conf_independent(100.0) == V.conf_independent(100.0, lig["num_tors"])
    # the torsion tree: model::set on random conformations, bit for bit
    rng = np.random.RandomState(0)
    for conf in RC.random_confs(rng, lig["conf0"], 6):
        assert np.array_equal(s.set_conf(conf), V.set_conf(c.h, conf)[0])


def test_long_ligand_spans_two_beads_and_bond_lists_follow_bead_order(chain):
    """model::assign_bonds visits candidates bead by bead (15 A); bonded_to()'s depth-first walk depends on it."""
    xyz = chain.lig["coords0"]
    assert np.linalg.norm(xyz[-3] - xyz[0]) > 15.0
    n_rec = chain.ref.n_grid_atoms
    # our pairs already equal the reference's (previous test); also check a bond list crossing the bead boundary
    far = int(np.argmax(np.linalg.norm(xyz - xyz[0], axis=1) > 15.0))
    assert sorted(chain.ref.bonds(n_rec + far)) == sorted(chain.ref.bonds(n_rec + far))
    assert len(chain.ref.bonds(n_rec + far)) >= 2


def test_truncated_inputs_behave_like_the_reference(capi, rigid_text):
    """EOF inside a BRANCH: the reference keeps what it read, unless the branch never met its second ("immobile")
    atom -- then VINA_CHECK(immobile_atom) fires (parsing.h:186,197).  Ours used to read out of bounds there."""
    flex = open(RC.FLEX_RES).read().splitlines()
    k = max(i for i, l in enumerate(flex) if l.startswith("BRANCH  13  22"))
    bad = "\n".join(flex[:k + 1] + [flex[k + 3]]) + "\n"    # BRANCH 13 22, then atom 23 and the end: atom 22 never seen
    with pytest.raises(ref.RefError):
        ref.Scene(rigid_text, None, flex_text=bad)
    with pytest.raises(capi.MiGninaError, match="end of file"):
        capi.read_pdbqt_receptor_flex(rigid_text, bad, is_text=True)
    for upto in (k + 1, k + 2, k + 5):                       # an empty branch; atom 22 only; inside a sub-branch
        ok = "\n".join(flex[:upto]) + "\n"                   # no ENDBRANCH, no END_RES
        s = ref.Scene(rigid_text, None, flex_text=ok)
        rows_xyz, rows_smt, n_mov, n_inflex = capi.read_pdbqt_receptor_flex(rigid_text, ok, is_text=True)
        xyz, smt, _ = s.atoms()
        assert n_mov == s.n_movable and n_mov + n_inflex == s.n_atoms
        assert np.array_equal(rows_xyz[:s.n_atoms], xyz) and np.array_equal(rows_smt[:s.n_atoms], smt)
    lig = RC.cys_adduct_ligand().splitlines()
    k = max(i for i, l in enumerate(lig) if l.startswith("BRANCH  13  22"))
    lig_bad = "\n".join(lig[:k + 1] + [lig[k + 3], "TORSDOF 10"]) + "\n"
    with pytest.raises(ref.RefError):
        ref.Scene(rigid_text, lig_bad)
    with pytest.raises(capi.MiGninaError):
        capi.read_pdbqt_ligand(lig_bad, is_text=True)
    with pytest.raises(ref.RefError):                        # a ligand without TORSDOF
        ref.Scene(rigid_text, "\n".join(lig[:k + 2]) + "\n")
    with pytest.raises(capi.MiGninaError, match="TORSDOF"):
        capi.read_pdbqt_ligand("\n".join(lig[:k + 2]) + "\n", is_text=True)


def test_flexible_receptor_rows_are_the_references(capi, rigid_text):
    flex = open(RC.FLEX_RES).read()
    s = ref.Scene(rigid_text, None, flex_text=flex)
    xyz, smt, _ = s.atoms()                               # movable, then inflex
    gx, gs = s.grid_atoms()
    rows_xyz, rows_smt, n_mov, n_inflex = capi.read_pdbqt_receptor_flex(rigid_text, flex, is_text=True)
    assert n_mov == s.n_movable and n_mov + n_inflex == s.n_atoms
    assert np.array_equal(rows_xyz[:s.n_atoms], xyz) and np.array_equal(rows_smt[:s.n_atoms], smt)
    assert np.array_equal(rows_xyz[s.n_atoms:], gx) and np.array_equal(rows_smt[s.n_atoms:], gs)


def test_pair_tables_and_exact_terms_bit_exact(adduct):
    s, T = adduct.ref, adduct.T
    rng = np.random.RandomState(5)
    r2 = np.concatenate([np.arange(0, 2049) / 32.0, rng.uniform(0, 64, 400)]).astype(np.float32)
    for t1 in range(0, 28, 3):
        for t2 in range(t1, 28, 2):
            fast, e, dor = s.table_eval(t1, t2, r2)
            fo = np.array([T.eval_fast(t1, t2, float(x)) for x in r2], dtype=np.float32)
            ed = np.array([T.eval_deriv(t1, t2, float(x)) for x in r2], dtype=np.float32)
            assert np.array_equal(fast, fo) and np.array_equal(e, ed[:, 0]) and np.array_equal(dor, ed[:, 1])
    for t1 in range(28):
        for t2 in range(t1, 28, 3):
            for r in np.linspace(0.4, 8.0, 23):
                assert s.pair_energy(t1, t2, float(r)) == V.pair_energy(t1, t2, float(r))


def _lattice(begin, end, n, idx):
    return np.stack([begin[i] + (end[i] - begin[i]) * idx[:, i].astype(np.float32) / np.float32(n[i])
                     for i in range(3)], 1).astype(np.float32)


def test_cache_grids_bit_exact_and_receptor_hydrogens_do_not_count(adduct):
    c = adduct
    rng = np.random.RandomState(1)
    for t, g in c.grids.items():
        idx = rng.randint(0, [c.n[0] + 1, c.n[1] + 1, c.n[2] + 1], size=(400, 3))
        probe = c.ref.cache_probe(t, _lattice(c.begin, c.end, c.n, idx), v=1e10)
        assert np.array_equal(probe, g[idx[:, 2], idx[:, 1], idx[:, 0]])
    # with the hydrogens in (round 1's reading of cache.cpp) the grids are visibly different
    t = 2
    wrong = V.cache_populate(c.T, c.gd, c.rec_xyz, np.where(c.rec_smt <= 1, 2, c.rec_smt).astype(np.int32), t)
    assert np.abs(wrong - c.grids[t]).max() > 0.5


def test_cache_grid_on_a_box_aligned_with_szv_grids_cells(capi, rigid_text):
    """Box begin = a multiple of 3: every 8th lattice plane starts a 3 A cell whose candidate list szv_grid_cache::get
    builds from a degenerate brick; atoms beyond 8 A of that plane are missing for the whole cell."""
    center, size = np.array([-7.5, 9.0, 0.7], np.float32), np.array([15.0, 16.0, 14.0], np.float32)
    c = Case(capi, rigid_text, RC.cys_adduct_ligand(), center, size)
    assert c.begin[0] == -15.0
    rng = np.random.RandomState(2)
    t = 2
    idx = rng.randint(0, [c.n[0] + 1, c.n[1] + 1, c.n[2] + 1], size=(3000, 3))
    probe = c.ref.cache_probe(t, _lattice(c.begin, c.end, c.n, idx), v=1e10)
    mine = c.grids[t][idx[:, 2], idx[:, 1], idx[:, 0]]
    assert np.array_equal(probe, mine)
    # and that really is the quirk: plain "every atom within 8 A" gives other numbers on this box
    full = np.zeros(len(idx), dtype=np.float32)
    pts = _lattice(c.begin, c.end, c.n, idx)
    heavy = c.rec_smt > 1
    for k in range(0, len(idx), 500):
        p = pts[k]
        r2 = ((c.rec_xyz[heavy] - p) ** 2).sum(1)
        acc = np.float32(0)
        for j in np.nonzero(r2 <= 64)[0]:
            acc = np.float32(acc + np.float32(c.T.eval_fast(int(c.rec_smt[heavy][j]), t, float(r2[j]))))
        full[k] = acc
    ks = np.arange(0, len(idx), 500)
    assert np.abs(full[ks] - mine[ks]).max() > 1e-3


@pytest.mark.parametrize("which", ["adduct", "chain"])
def test_eval_and_eval_deriv_bit_exact(which, request):
    c = request.getfixturevalue(which)
    s, lig = c.ref, c.lig
    rng = np.random.RandomState(3)
    confs = np.concatenate([RC.random_confs(rng, lig["conf0"], 4, small=True), RC.random_confs(rng, lig["conf0"], 4),
                            RC.random_confs(rng, lig["conf0"], 2, spread=9.0)])          # the last two leave the box
    for conf in confs:
        for v in (V3, HUNT):
            er, cr, xr, fr = s.eval_deriv(conf, v)
            eo, co, xo, fo = c.ora.eval_deriv(conf, v)
            assert er == eo and np.array_equal(cr, co) and np.array_equal(xr, xo)
            assert np.array_equal(np.abs(fr), np.abs(fo))
            assert s.eval(conf, v) == c.ora.eval(conf, v)                               # model::eval
            assert s.ig_eval(conf, v[1]) == V.cache_eval(c.ora, conf, v[1])            # cache::eval (Metropolis)
        er, cr, _, _ = s.eval_deriv(conf, V3, ig=1)                                     # non_cache
        eo, co, _, _ = V.noncache_eval(c.ora, c.rec_xyz, c.rec_smt, conf, V3)
        assert er == eo and np.array_equal(cr, co)
        assert s.eval(conf, V3, ig=1) == V.noncache_eval(c.ora, c.rec_xyz, c.rec_smt, conf, V3, deriv=False)[0]
        assert s.within(conf) == bool(V._voxel.lib().ora_vina_within(
            V.C.byref(c.gd), V.C.byref(c.h.c), conf.ctypes.data_as(V.C.POINTER(V.C.c_float))))


def test_final_energies_follow_do_searchs_docking_branch(adduct):
    """main.cpp:231,339-344: receptor term from a non_cache on the LINEAR tables, pair terms exact, num_tors_div."""
    c = adduct
    rng = np.random.RandomState(4)
    for conf in RC.random_confs(rng, c.lig["conf0"], 5, small=True, spread=0.5):
        ef, intra = c.ref.final_energies(conf)
        _, _, inter_lin, _ = V.noncache_eval(c.ora, c.rec_xyz, c.rec_smt, conf, V3, deriv=False, exact=False)
        _, _, _, intra_x = V.noncache_eval(c.ora, c.rec_xyz, c.rec_smt, conf, V3, deriv=False, exact=True)
        assert intra == intra_x
        mine = V.conf_independent(np.float32(np.float32(inter_lin) + np.float32(intra_x)) - np.float32(intra_x),
                                  c.lig["num_tors"])
        assert ef == mine


@pytest.mark.parametrize("which", ["adduct", "chain"])
def test_bfgs_runs_are_bit_identical(which, request):
    c = request.getfixturevalue(which)
    rng = np.random.RandomState(6)
    confs = np.concatenate([RC.random_confs(rng, c.lig["conf0"], 3, small=True), RC.random_confs(rng, c.lig["conf0"], 3)])
    for conf in confs:
        g = rng.normal(size=c.ref.change_len).astype(np.float32)
        assert np.array_equal(c.ref.conf_increment(conf, g, 0.37), V.conf_increment(conf, g, 0.37, c.ref.n_lig_tors))
        for v in (V3, HUNT):
            for iters in (1, 3, c.max_iters):
                er, xr, gr = c.ref.bfgs(conf, v, max_iters=iters)
                eo, xo, go, _ = c.ora.bfgs(conf, v, max_iters=iters)
                assert er == eo and np.array_equal(xr, xo) and np.array_equal(gr, go)


def test_random_distributions_of_the_stand_in_boost(adduct):
    """oracle kind-1 generator == what oracle/ref_shims/boost/random.hpp hands the reference (mt19937 is standard;
    the distributions are restatements: the stream of a real Boost build is unpinned)."""
    for seed in (0, 7, 12345):
        a, b = ref.random_stream(seed, 200), V.random_stream(1, seed, 200)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert a[0].min() >= 0 and a[0].max() < 1 and set(a[1]) == set(range(10)) and abs(a[2].mean()) < 0.3
    for seed in range(8):
        x = adduct.ref.mutate(adduct.lig["conf0"], seed)
        assert np.array_equal(x, V.mutate(adduct.h, adduct.lig["conf0"], seed))


@pytest.mark.parametrize("which,steps", [("adduct", 1), ("adduct", 40), ("adduct", 600), ("chain", 150)])
def test_monte_carlo_chains_are_bit_identical(which, steps, request):
    """monte_carlo::operator() (mutate -> BFGS(hunt) -> Metropolis on what `model` holds -> BFGS(full) -> container)
    on the same mt19937 stream: same containers, energies, conformations and coordinates, bit for bit."""
    c = request.getfixturevalue(which)
    for seed in (1, 2, 3):
        er, cr, xr = c.ref.mc(seed, steps, c.begin, c.end, max_iters=c.max_iters, num_saved=20)
        eo, co, xo, _ = V.mc_chain(c.ora, c.begin, c.end, seed, steps, c.max_iters, num_saved=20, rng_kind=1,
                                   conf0=c.lig["conf0"])
        assert len(er) == len(eo) >= 1
        assert np.array_equal(er, eo) and np.array_equal(cr, co) and np.array_equal(xr, xo)


# ---- flexible residues in the search: model = rigid + flex + ligand ----------------------------
@pytest.fixture(scope="module")
def flexcase(capi, rigid_text):
    flex = open(RC.FLEX_RES).read()
    lig_txt = RC.long_chain_ligand(n=8, origin=(-9.0, 12.0, 3.0))
    s = ref.Scene(rigid_text, lig_txt, flex_text=flex)
    rx, rs, d = capi.read_pdbqt_model(rigid_text, flex, lig_txt, is_text=True)
    center, size = RC.box_of(d["coords0"][:d["n_movable"]])
    b, e, n = s.build_grids(center, size)
    gd = V.setup_grid_dims(center, size)
    T = V.Tables()
    types = sorted(set(int(t) for t in d["smt"][:d["n_movable"]] if t > 1))
    grids = {t: V.cache_populate(T, gd, rx, rs, t) for t in types}
    h = V.LigandHandle(d)
    return dict(ref=s, d=d, rx=rx, rs=rs, b=b, e=e, ora=V.Scene(T, gd, grids, h), h=h)


def test_flex_model_is_the_references(flexcase):
    """parse_receptor_pdbqt(rigid, flex) + m.append(ligand): atom order [flex movable | ligand | inflex], types of the
    combined receptor model, model::other_pairs (initialize_pairs + every receptor-model / ligand atom pair of
    model::append, hydrogens included) and the ligand's pairs in the reference's order, conf = [7 + T_lig + T_flex]."""
    s, d = flexcase["ref"], flexcase["d"]
    xyz, smt, _ = s.atoms()
    assert s.n_flex == 1 and s.n_flex_tors == 10 == d["n_flex_tors"] and s.n_lig_tors == d["n_lig_tors"]
    assert np.array_equal(smt, d["smt"]) and np.array_equal(xyz, d["coords0"]) and s.n_movable == d["n_movable"]
    assert (s.lig_begin, s.lig_end) == (d["lig_begin"], d["lig_end"])
    gx, gs = s.grid_atoms()
    assert np.array_equal(gs, flexcase["rs"]) and np.array_equal(gx, flexcase["rx"])
    assert np.array_equal(s.pairs(other=True)[0], d["pairs"][d["pair_kind"] == 1])
    assert np.array_equal(s.pairs()[0], d["pairs"][d["pair_kind"] == 0])
    assert np.array_equal(s.initial_conf(), d["conf0"])
    rng = np.random.RandomState(0)
    for conf in RC.random_confs(rng, d["conf0"], 5):
        assert np.array_equal(s.set_conf(conf), V.set_conf(flexcase["h"], conf)[0])


def test_flex_energies_bfgs_and_monte_carlo_bit_identical(flexcase):
    """flex.derivative (tree.h:374-393), other_pairs with the v[2] cap (model.cu:209-213), the receptor term of the
    movable side-chain atoms, BFGS over 6 + T_lig + T_flex variables, mutate_conf over ligand AND residue torsions."""
    s, d, ora = flexcase["ref"], flexcase["d"], flexcase["ora"]
    rng = np.random.RandomState(1)
    mi = (25 + s.n_movable) // 3
    for conf in np.concatenate([RC.random_confs(rng, d["conf0"], 3, small=True), RC.random_confs(rng, d["conf0"], 2)]):
        for v in (V3, HUNT):
            er, cr, xr, fr = s.eval_deriv(conf, v)
            eo, co, xo, fo = ora.eval_deriv(conf, v)
            assert er == eo and np.array_equal(cr, co) and np.array_equal(xr, xo)
            assert s.eval(conf, v) == ora.eval(conf, v) and s.ig_eval(conf, v[1]) == V.cache_eval(ora, conf, v[1])
            er, xr, gr = s.bfgs(conf, v, max_iters=mi)
            eo, xo, go, _ = ora.bfgs(conf, v, max_iters=mi)
            assert er == eo and np.array_equal(xr, xo) and np.array_equal(gr, go)
        er, cr, _, _ = s.eval_deriv(conf, V3, ig=1)
        eo, co, _, _ = V.noncache_eval(ora, flexcase["rx"], flexcase["rs"], conf, V3)
        assert er == eo and np.array_equal(cr, co)
    for seed, steps in ((1, 5), (2, 80), (3, 400)):
        er, cr, xr = s.mc(seed, steps, flexcase["b"], flexcase["e"], max_iters=mi, num_saved=20)
        eo, co, xo, _ = V.mc_chain(ora, flexcase["b"], flexcase["e"], seed, steps, mi, num_saved=20, rng_kind=1,
                                   conf0=d["conf0"])
        assert np.array_equal(er, eo) and np.array_equal(cr, co) and np.array_equal(xr, xo)


def _setup_user_gd(center, nelem, spacing):
    """main.cpp:635-670 in its own types (fl = float, atof = double)"""
    g = np.float32(spacing)
    size = [np.float32((float(k) + 1) * float(g)) for k in nelem]
    ctr = [np.float32(float("%.3f" % c) + 0.5 * float(g)) for c in center]
    n = [int(np.ceil(np.float32(s / g))) for s in size]
    begin = [np.float32(c - np.float32(g * np.float32(k)) / np.float32(2)) for c, k in zip(ctr, n)]
    end = [np.float32(b + np.float32(g * np.float32(k))) for b, k in zip(begin, n)]
    return np.array(begin, np.float32), np.array(end, np.float32), np.array(n, np.int32)


def test_user_grid_goes_into_the_cache_and_the_non_cache_derivative_like_the_reference(capi, rigid_text):
    """--user_grid (main.cpp:1342-1350): cache::populate adds evaluate_user at every lattice point with the point's
    INDICES as the location (cache.cpp:177-179); non_cache::eval_deriv adds it per heavy atom at the atom's coordinates
    before the curl (non_cache.cpp:168-173); non_cache::eval does not see it, model::eval adds its own sum over all ligand atoms (model.cu:125-134).  Reader, restatement
    and reference on the same file."""
    lig_text = RC.cys_adduct_ligand()
    lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
    center, size = RC.box_of(lig["coords0"])
    text, value_lines = RC.user_grid_text(center, (18, 20, 16), 0.75)
    b, e, n, vals = capi.user_grid_parse(text)
    b0, e0, n0 = _setup_user_gd(center, (18, 20, 16), 0.75)
    assert np.array_equal(n, n0) and np.array_equal(b, b0) and np.array_equal(e, e0)
    assert vals.shape == (n[2], n[1], n[0]) and vals.ravel()[7] == float(value_lines.split("\n")[7])
    scale = np.float32(1 - 0.25)                                  # --user_grid_lambda 0.25
    plain = Case(capi, rigid_text, lig_text)                     # the same scene without the user grid
    try:
        s = ref.Scene(rigid_text, lig_text)
        s.set_user_grid(b, e, n, value_lines, scale)
        begin, end, nn = s.build_grids(center, size)
        V.set_user_grid(b, e, n, vals, scale)
        T, gd = V.Tables(), V.setup_grid_dims(center, size)
        rec_xyz, rec_smt = s.grid_atoms()
        types = sorted(set(int(t) for t in lig["smt"] if t > 1))
        grids = {t: V.cache_populate(T, gd, rec_xyz, rec_smt, t) for t in types}
        rng = np.random.RandomState(4)
        for t in types[:2]:
            idx = rng.randint(0, [nn[0] + 1, nn[1] + 1, nn[2] + 1], size=(500, 3))
            probe = s.cache_probe(t, _lattice(begin, end, nn, idx), v=3.4e38)   # not_max(v) false: no curl of the big values
            assert np.array_equal(probe, grids[t][idx[:, 2], idx[:, 1], idx[:, 0]])
        assert np.abs(grids[types[0]] - plain.grids[types[0]]).max() > 0.1
        ora = V.Scene(T, gd, grids, V.LigandHandle(lig))
        for conf in RC.random_confs(np.random.RandomState(5), lig["conf0"], 6):
            er, cr, _, _ = s.eval_deriv(conf, V3, ig=1)         # non_cache::eval_deriv sees the user grid ...
            eo, co, _, _ = V.noncache_eval(ora, rec_xyz, rec_smt, conf, V3)
            assert er == eo and np.array_equal(cr, co)
            ep, _, _, _ = plain.ref.eval_deriv(conf, V3, ig=1)
            assert er != ep
            # ... non_cache::eval does not (non_cache.cpp:76), but model::eval adds its own term over ALL ligand atoms
            # (model.cu:125-134): that is how it reaches eval_adjusted / the final energies
            assert s.ig_eval(conf, 1000.0, ig=1) == plain.ref.ig_eval(conf, 1000.0, ig=1)
            assert s.eval(conf, V3, ig=1) != plain.ref.eval(conf, V3, ig=1)
            assert s.eval(conf, V3, ig=1) == V.noncache_eval(ora, rec_xyz, rec_smt, conf, V3, deriv=False)[0]
            assert s.eval(conf, V3) == ora.eval(conf, V3)
            er, cr, _, _ = s.eval_deriv(conf, V3)                # cache: through the baked lattice
            eo, co, _, _ = ora.eval_deriv(conf, V3)
            assert er == eo and np.array_equal(cr, co)
    finally:
        V.set_user_grid()


def test_spline_approximation_follows_the_reference(capi, rigid_text):
    """--approximation spline = precalculate_splines(sf, factor) (precalculate.h:277-449, splines.h), gnina's default
    for --minimize with factor 10 (main.cpp:1162-1165).  The reference builds each spline by inverting a dense fp32
    matrix with Eigen; the restatement solves the same (tridiagonal) system directly -- so: close, not bit-identical
    (spline values and derivatives to ~1e-6 of their scale), and everything built on them within the usual bars."""
    lig_text = RC.cys_adduct_ligand()
    lig = capi.read_pdbqt_ligand(lig_text, is_text=True)
    center, size = RC.box_of(lig["coords0"])
    s = ref.Scene(rigid_text, lig_text)
    s.set_approximation(1, 10.0)
    try:
        V.set_approximation(1, 10.0)
        T = V.Tables()
        r2 = np.concatenate([np.linspace(0.05, 63.9, 400), [63.99, 64.0, 70.0]]).astype(np.float32)
        for t1, t2 in ((2, 2), (2, 13), (7, 13), (10, 4), (3, 8)):
            e0, d0 = s.prec_eval(t1, t2, r2)
            e1 = np.array([T.eval_fast(t1, t2, float(x)) for x in r2], np.float32)
            e2, d2 = np.zeros_like(r2), np.zeros_like(r2)
            for k, x in enumerate(r2):
                e2[k], d2[k] = T.eval_deriv(t1, t2, float(x))
            scale = max(1e-3, np.abs(e0).max())
            assert np.abs(e1 - e0).max() <= 2e-5 * scale and np.abs(e2 - e0).max() <= 2e-5 * scale, (t1, t2)
            assert np.abs(d2 - d0).max() <= 2e-4 * max(1e-3, np.abs(d0).max()), (t1, t2)
        begin, end, n = s.build_grids(center, size)
        gd = V.setup_grid_dims(center, size)
        rec_xyz, rec_smt = s.grid_atoms()
        types = sorted(set(int(t) for t in lig["smt"] if t > 1))
        grids = {t: V.cache_populate(T, gd, rec_xyz, rec_smt, t) for t in types}
        rng = np.random.RandomState(6)
        t = types[0]
        idx = rng.randint(0, [n[0] + 1, n[1] + 1, n[2] + 1], size=(400, 3))
        probe = s.cache_probe(t, _lattice(begin, end, n, idx), v=3.4e38)
        mine = grids[t][idx[:, 2], idx[:, 1], idx[:, 0]]
        assert np.abs(mine - probe).max() <= 1e-5 * max(1.0, np.abs(probe).max())
        ora = V.Scene(T, gd, grids, V.LigandHandle(lig))
        for conf in RC.random_confs(np.random.RandomState(7), lig["conf0"], 5):
            er, cr, _, _ = s.eval_deriv(conf, V3)
            eo, co, _, _ = ora.eval_deriv(conf, V3)
            assert abs(er - eo) <= 1e-5 * max(1.0, abs(er)) and np.abs(cr - co).max() <= 1e-4 * max(1.0, np.abs(cr).max())
            er, cr, _, _ = s.eval_deriv(conf, V3, ig=1)
            eo, co, _, _ = V.noncache_eval(ora, rec_xyz, rec_smt, conf, V3)
            assert abs(er - eo) <= 1e-5 * max(1.0, abs(er)) and np.abs(cr - co).max() <= 1e-4 * max(1.0, np.abs(cr).max())
    finally:
        V.set_approximation(0)


@pytest.mark.parametrize("which", ["adduct", "chain"])
def test_accurate_line_search_runs_are_bit_identical(which, request):
    """--accurate_line_search (bfgs.h:104-180, compute_lambdamin :93-102): whole quasi_newton runs and Monte-Carlo chains
    of the restatement against the reference, on the cache and on non_cache."""
    c = request.getfixturevalue(which)
    s, lig = c.ref, c.lig
    rng = np.random.RandomState(8)
    confs = np.concatenate([RC.random_confs(rng, lig["conf0"], 3, small=True), RC.random_confs(rng, lig["conf0"], 3)])
    try:
        s.set_line_search(True)
        V.set_line_search(True)
        differs = 0
        for conf in confs:
            for v in (V3, HUNT):
                for iters in (1, 4, c.max_iters):
                    er, xr, gr = s.bfgs(conf, v, max_iters=iters)
                    eo, xo, go, _ = c.ora.bfgs(conf, v, max_iters=iters)
                    assert er == eo and np.array_equal(xr, xo) and np.array_equal(gr, go), (v, iters)
            s.set_line_search(False)
            ef, xf, _ = s.bfgs(conf, V3, max_iters=c.max_iters)
            s.set_line_search(True)
            ea, xa, _ = s.bfgs(conf, V3, max_iters=c.max_iters)
            differs += int(not np.array_equal(xf, xa))
        assert differs >= 3                                          # it really is another search
        for seed, steps in ((1, 30), (2, 120)):
            er, cr, xr = s.mc(seed, steps, c.begin, c.end, max_iters=c.max_iters, num_saved=20)
            eo, co, xo, _ = V.mc_chain(c.ora, c.begin, c.end, seed, steps, c.max_iters, num_saved=20)
            assert len(er) == len(eo) and np.array_equal(er, eo) and np.array_equal(cr, co) and np.array_equal(xr, xo)
    finally:
        s.set_line_search(False)
        V.set_line_search(False)


@pytest.mark.parametrize("which", ["adduct", "chain"])
def test_simple_ascent_runs_are_bit_identical(which, request):
    """--simple_ascent = minimization_params::Simple (quasi_newton.cpp:77-79): simple_gradient_ascent (bfgs.h:234-355),
    steepest descent under the accurate line search, no quasi-Newton update -- restatement vs the reference over whole
    minimisations and Monte-Carlo chains; and it is neither of the two bfgs<> variants."""
    c = request.getfixturevalue(which)
    s, lig = c.ref, c.lig
    rng = np.random.RandomState(9)
    confs = np.concatenate([RC.random_confs(rng, lig["conf0"], 3, small=True), RC.random_confs(rng, lig["conf0"], 2)])
    try:
        differs = 0
        for conf in confs:
            s.set_line_search(False, simple=True)
            V.set_line_search(False, simple=True)
            for v in (V3, HUNT):
                for iters in (1, 4, c.max_iters):
                    er, xr, gr = s.bfgs(conf, v, max_iters=iters)
                    eo, xo, go, _ = c.ora.bfgs(conf, v, max_iters=iters)
                    assert er == eo and np.array_equal(xr, xo) and np.array_equal(gr, go), (v, iters)
            es, xs, _ = s.bfgs(conf, V3, max_iters=c.max_iters)
            s.set_line_search(True)
            ea, xa, _ = s.bfgs(conf, V3, max_iters=c.max_iters)
            differs += int(not np.array_equal(xs, xa))
        assert differs >= 3
        s.set_line_search(False, simple=True)
        V.set_line_search(False, simple=True)
        er, cr, xr = s.mc(3, 60, c.begin, c.end, max_iters=c.max_iters, num_saved=20)
        eo, co, xo, _ = V.mc_chain(c.ora, c.begin, c.end, 3, 60, c.max_iters, num_saved=20)
        assert len(er) == len(eo) and np.array_equal(er, eo) and np.array_equal(cr, co) and np.array_equal(xr, xo)
    finally:
        s.set_line_search(False)
        V.set_line_search(False)


def test_final_energies_with_flexible_residues(flexcase, adduct):
    """main.cpp:339-344 on the combined model: eval_intramolecular (model.cu:352-399) = ligand pairs + flexible atoms
    against the rigid receptor (every pair curled with v[1], exact tables) + other_pairs without a ligand atom;
    e = model::eval(exact_prec, non_cache) = non_cache::eval (all movable atoms, linear tables) + other_pairs + ligand
    pairs; reported = conf_independent(e - intramolecular).  Restatement vs the reference, and the rigid case as before."""
    s, d, ora = flexcase["ref"], flexcase["d"], flexcase["ora"]
    rng = np.random.RandomState(9)
    nt = s.conf_independent(100.0)
    for conf in np.concatenate([RC.random_confs(rng, d["conf0"], 4, small=True), RC.random_confs(rng, d["conf0"], 2)]):
        ef, intra = s.final_energies(conf)
        e, intra_o = V.model_energies(ora, flexcase["rx"], flexcase["rs"], conf, V3)
        assert intra == intra_o
        x = np.float32(np.float32(e) - np.float32(intra_o))
        assert ef == s.conf_independent(float(x))
    assert nt == s.conf_independent(100.0)
    c = adduct                                              # a rigid receptor through the same function
    for conf in RC.random_confs(rng, c.lig["conf0"], 3, small=True, spread=0.5):
        ef, intra = c.ref.final_energies(conf)
        e, intra_o = V.model_energies(c.ora, c.rec_xyz, c.rec_smt, conf, V3)
        assert intra == intra_o and ef == V.conf_independent(np.float32(np.float32(e) - np.float32(intra_o)), c.lig["num_tors"])
