"""oracle/vina_ref.c and the native PDBQT reader against the frozen outputs of the reference's own Vina code
(tests/golden/vina_goldens.npz, made from oracle/_ref by tests/golden/make_vina_goldens.py).  Unlike
tests/test_ref_vina.py this needs neither /root/reference nor oracle/_ref: it runs anywhere.  Bar: bit-exact."""
import os

import numpy as np
import pytest

from oracle import vina as V

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vina_goldens.npz"))
V3, HUNT = (1000.0, 1000.0, 1000.0), (10.0, 10.0, 10.0)
CASES = ["adduct", "chain", "aligned"]


@pytest.fixture(scope="module")
def capi():
    from gnina_amd import build, capi as c
    build.build()
    return c


_cache = {}


def scene(capi, name):
    if name not in _cache:
        P = name + "/"
        lig = capi.read_pdbqt_ligand(bytes(G[P + "lig_text"]).decode(), is_text=True)
        gd = V.setup_grid_dims(G[P + "center"], G[P + "size"])
        T = V.Tables()
        h = V.LigandHandle(lig)
        grids = {int(t): V.cache_populate(T, gd, G[P + "rec_xyz"], G[P + "rec_smt"], int(t)) for t in G[P + "types"]}
        _cache[name] = (lig, gd, T, h, grids, V.Scene(T, gd, grids, h))
    return _cache[name]


@pytest.mark.parametrize("name", CASES)
def test_reader_and_box(capi, name):
    P = name + "/"
    lig, gd, *_ = scene(capi, name)
    assert np.array_equal(lig["smt"], G[P + "atoms_smt"]) and np.array_equal(lig["coords0"], G[P + "atoms_xyz"])
    assert np.array_equal(lig["pairs"], G[P + "pairs"]) and np.array_equal(lig["conf0"], G[P + "conf0"])
    assert V.conf_independent(100.0, lig["num_tors"]) == G[P + "num_tors_div_of_100"]
    assert np.array_equal(np.float32(gd.begin[:]), G[P + "begin"]) and list(gd.n) == list(G[P + "n"])


def test_tables_and_random_stream():
    T = V.Tables()
    r2 = G["tables/r2"]
    for k, (a, b) in enumerate(G["tables/pairs"]):
        f = np.array([T.eval_fast(int(a), int(b), float(x)) for x in r2], np.float32)
        ed = np.array([T.eval_deriv(int(a), int(b), float(x)) for x in r2], np.float32)
        assert np.array_equal(f, G["tables/fast"][k]) and np.array_equal(ed[:, 0], G["tables/e"][k])
        assert np.array_equal(ed[:, 1], G["tables/dor"][k])
    u, i, g = V.random_stream(1, 12345, 64)
    assert np.array_equal(u, G["rng/uniform01"]) and np.array_equal(i, G["rng/int_0_9"]) and np.array_equal(g, G["rng/normal"])


@pytest.mark.parametrize("name", CASES)
def test_grids_eval_and_final_energies(capi, name):
    P = name + "/"
    lig, gd, T, h, grids, S = scene(capi, name)
    idx = G[P + "grid_idx"]
    for k, t in enumerate(G[P + "types"]):
        assert np.array_equal(grids[int(t)][idx[:, 2], idx[:, 1], idx[:, 0]], G[P + "grid_val"][k])
    rx, rs = G[P + "rec_xyz"], G[P + "rec_smt"]
    for i, conf in enumerate(G[P + "confs"]):
        for tag, v in (("v1000", V3), ("v10", HUNT)):
            e, ch, xyz, _ = S.eval_deriv(conf, v)
            assert e == G[P + tag + "/e"][i] and np.array_equal(ch, G[P + tag + "/change"][i])
            assert np.array_equal(xyz, G[P + tag + "/coords"][i])
            assert S.eval(conf, v) == G[P + tag + "/eval"][i]
            assert V.cache_eval(S, conf, v[1]) == G[P + tag + "/ig_eval"][i]
        e, ch, inter, intra = V.noncache_eval(S, rx, rs, conf, V3)
        assert e == G[P + "noncache/e"][i] and np.array_equal(ch, G[P + "noncache/change"][i])
        tot, _, inter_l, _ = V.noncache_eval(S, rx, rs, conf, V3, deriv=False)
        assert tot == G[P + "noncache/eval"][i] and inter_l == G[P + "noncache/ig_eval"][i]
        _, _, _, intra_x = V.noncache_eval(S, rx, rs, conf, V3, deriv=False, exact=True)
        assert intra_x == G[P + "final/intra"][i]
        ef = V.conf_independent(np.float32(np.float32(inter_l) + np.float32(intra_x)) - np.float32(intra_x), lig["num_tors"])
        assert ef == G[P + "final/e"][i]


@pytest.mark.parametrize("name", CASES[:2])
def test_bfgs_and_monte_carlo(capi, name):
    P = name + "/"
    lig, gd, T, h, grids, S = scene(capi, name)
    mi = int(G[P + "max_iters"])
    for tag, v in (("v1000", V3), ("v10", HUNT)):
        for iters in (1, 3, mi):
            for i, conf in enumerate(G[P + "confs"][:12]):
                e, x, g, _ = S.bfgs(conf, v, max_iters=iters)
                assert e == G[P + f"bfgs/{tag}/{iters}/e"][i] and np.array_equal(x, G[P + f"bfgs/{tag}/{iters}/conf"][i])
                assert np.array_equal(g, G[P + f"bfgs/{tag}/{iters}/grad"][i])
    for key in [k for k in G.files if k.startswith(P + "mc/") and k.endswith("/e")]:
        seed, steps = (int(x) for x in key.split("/")[2].split("_"))
        e, cf, xyz, _ = V.mc_chain(S, G[P + "begin"], G[P + "end"], seed, steps, mi, num_saved=20, rng_kind=1,
                                   conf0=lig["conf0"])
        base = key[:-2]
        assert np.array_equal(e, G[base + "/e"]) and np.array_equal(cf, G[base + "/conf"])
        assert np.array_equal(xyz, G[base + "/coords"])


def _flex_desc():
    P = "flex/desc/"
    d = {k: G[P + k] for k in ("smt", "local_xyz", "parent", "abeg", "aend", "rel_origin", "rel_axis", "pairs",
                               "pair_kind", "conf0")}
    d["n_movable"], d["lig_begin"], d["lig_end"] = (int(x) for x in G[P + "ints"])
    d["n_tors"] = len(d["parent"]) - 1
    return d


def test_flexible_residues_in_the_search():
    """model = rigid + flexible side chain + ligand: eval_deriv / eval / cache::eval / non_cache,
    whole BFGS runs over 6 + T_ligand + T_flex variables and a Monte-Carlo chain, against the frozen reference."""
    P = "flex/"
    d = _flex_desc()
    gd = V.setup_grid_dims(G[P + "center"], G[P + "size"])
    T = V.Tables()
    h = V.LigandHandle(d)
    rx, rs = G[P + "rec_xyz"], G[P + "rec_smt"]
    grids = {int(t): V.cache_populate(T, gd, rx, rs, int(t)) for t in G[P + "types"]}
    S = V.Scene(T, gd, grids, h)
    assert np.array_equal(d["conf0"], G[P + "conf0"])
    mi = int(G[P + "max_iters"])
    for i, conf in enumerate(G[P + "confs"]):
        for tag, v in (("v1000", V3), ("v10", HUNT)):
            e, ch, xyz, _ = S.eval_deriv(conf, v)
            assert e == G[P + tag + "/e"][i] and np.array_equal(ch, G[P + tag + "/change"][i])
            assert np.array_equal(xyz, G[P + tag + "/coords"][i])
            assert S.eval(conf, v) == G[P + tag + "/eval"][i] and V.cache_eval(S, conf, v[1]) == G[P + tag + "/ig_eval"][i]
        e, ch, _, _ = V.noncache_eval(S, rx, rs, conf, V3)
        assert e == G[P + "noncache/e"][i] and np.array_equal(ch, G[P + "noncache/change"][i])
        if i < 12:
            e, x, g, _ = S.bfgs(conf, HUNT, max_iters=mi)
            assert e == G[P + f"bfgs/v10/{mi}/e"][i] and np.array_equal(x, G[P + f"bfgs/v10/{mi}/conf"][i])
    e, cf, xyz, _ = V.mc_chain(S, G[P + "begin"], G[P + "end"], 1, 40, mi, num_saved=20, rng_kind=1, conf0=d["conf0"])
    assert np.array_equal(e, G[P + "mc/1_40/e"]) and np.array_equal(cf, G[P + "mc/1_40/conf"])
    assert np.array_equal(xyz, G[P + "mc/1_40/coords"])


def test_user_grid_restatement_against_the_frozen_reference(capi):
    """oracle/vina_ref.c's --user_grid terms against tests/golden/user_grid_goldens.npz (reference outputs)"""
    U = np.load(os.path.join(os.path.dirname(__file__), "golden", "user_grid_goldens.npz"))
    lig = capi.read_pdbqt_ligand(bytes(U["lig_text"]).decode(), is_text=True)
    ub, ue, un, vals = capi.user_grid_parse(bytes(U["user_grid_text"]))
    assert np.array_equal(ub, U["ub"]) and np.array_equal(ue, U["ue"]) and np.array_equal(un, U["un"])
    try:
        V.set_user_grid(ub, ue, un, vals, float(U["scale"]))
        T, gd = V.Tables(), V.setup_grid_dims(U["center"], U["size"])
        grids = {int(t): V.cache_populate(T, gd, U["rec_xyz"], U["rec_smt"], int(t)) for t in U["types"]}
        idx = U["grid_idx"]
        for k, t in enumerate(U["types"]):
            assert np.array_equal(grids[int(t)][idx[:, 2], idx[:, 1], idx[:, 0]], U["grid_val"][k])
        ora = V.Scene(T, gd, grids, V.LigandHandle(lig))
        for b, conf in enumerate(U["confs"]):
            e, ch, _, _ = V.noncache_eval(ora, U["rec_xyz"], U["rec_smt"], conf, (1000.0, 1000.0, 1000.0))
            assert e == U["noncache/e"][b] and np.array_equal(ch, U["noncache/change"][b])
            assert V.noncache_eval(ora, U["rec_xyz"], U["rec_smt"], conf, (1000.0, 1000.0, 1000.0), deriv=False)[0] == \
                U["noncache/eval"][b]
            e, ch, _, _ = ora.eval_deriv(conf, (1000.0, 1000.0, 1000.0))
            assert e == U["cache/e"][b] and np.array_equal(ch, U["cache/change"][b])
            assert ora.eval(conf, (1000.0, 1000.0, 1000.0)) == U["cache/eval"][b]
    finally:
        V.set_user_grid()
