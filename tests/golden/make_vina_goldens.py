#!/usr/bin/env python3
"""Freeze outputs of THE REFERENCE's own Vina code (oracle/_ref = gnina's sources compiled in place behind stand-in
headers, oracle/Makefile.ref) into a fixture the GPU tests can use where /root/reference does not exist.

Run in the build container:   python tests/golden/make_vina_goldens.py
Reads   /root/reference/test/gnina/data/GSK3B_DFG_out_35-388-processed_rigid.pdbqt, flex_res_side_chain.pdbqt
Writes  tests/golden/vina_goldens.npz: per case the inputs (typed receptor atoms, ligand PDBQT text, box, seeded
        conformations) and what the reference computes from them: parsed atoms / types / pairs, table samples,
        cache-grid samples, model::eval_deriv / eval / igrid::eval on cache and non_cache, final energies,
        quasi_newton results, monte_carlo containers (mt19937 + the stand-in Boost distributions).
Values only -- no reference source is copied."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402
from tests import ref_cases as RC  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vina_goldens.npz")
V3, HUNT = (1000.0, 1000.0, 1000.0), (10.0, 10.0, 10.0)


def lattice(b, e, n, idx):
    return np.stack([b[i] + (e[i] - b[i]) * idx[:, i].astype(np.float32) / np.float32(n[i]) for i in range(3)],
                    1).astype(np.float32)


def case(G, name, rigid, lig_text, center=None, size=None, seed=0, mc=((1, 60), (2, 200)), flex_text=None):
    s = ref.Scene(rigid, lig_text, flex_text=flex_text)
    xyz, smt, _ = s.atoms()
    rx, rs = s.grid_atoms()
    if center is None:
        center, size = RC.box_of(xyz[:s.n_movable])
    b, e, n = s.build_grids(center, size)
    rng = np.random.RandomState(seed)
    conf0 = s.initial_conf()
    P = name + "/"
    G[P + "lig_text"] = np.frombuffer(lig_text.encode(), dtype=np.uint8)
    if flex_text is not None:
        # the description our reader builds from (rigid, flex, ligand) -- checked against the reference's model in
        # tests/test_ref_vina.py -- so that the GPU test needs no reference file
        from gnina_amd import capi
        _, _, d = capi.read_pdbqt_model(rigid, flex_text, lig_text, is_text=True)
        for k in ("smt", "local_xyz", "parent", "abeg", "aend", "rel_origin", "rel_axis", "pairs", "pair_kind", "conf0"):
            G[P + "desc/" + k] = np.asarray(d[k])
        G[P + "desc/ints"] = np.array([d["n_movable"], d["lig_begin"], d["lig_end"]], np.int32)
    G[P + "rec_xyz"], G[P + "rec_smt"] = rx, rs
    G[P + "center"], G[P + "size"] = np.asarray(center, np.float32), np.asarray(size, np.float32)
    G[P + "begin"], G[P + "end"], G[P + "n"] = b, e, n
    G[P + "atoms_xyz"], G[P + "atoms_smt"], G[P + "pairs"] = xyz, smt, s.pairs()[0]
    G[P + "conf0"] = conf0
    G[P + "num_tors_div_of_100"] = np.float32(s.conf_independent(100.0))
    types = sorted(set(int(t) for t in smt[:s.n_movable] if t > 1))
    G[P + "types"] = np.array(types, np.int32)
    idx = rng.randint(0, [n[0] + 1, n[1] + 1, n[2] + 1], size=(600, 3)).astype(np.int32)
    G[P + "grid_idx"] = idx
    G[P + "grid_val"] = np.stack([s.cache_probe(t, lattice(b, e, n, idx), v=1e10) for t in types])
    confs = np.concatenate([RC.random_confs(rng, conf0, 6, small=True), RC.random_confs(rng, conf0, 6),
                            RC.random_confs(rng, conf0, 2, spread=9.0)])
    G[P + "confs"] = confs
    for tag, v in (("v1000", V3), ("v10", HUNT)):
        r = [s.eval_deriv(c, v) for c in confs]
        G[P + tag + "/e"] = np.array([x[0] for x in r], np.float32)
        G[P + tag + "/change"] = np.stack([x[1] for x in r])
        G[P + tag + "/coords"] = np.stack([x[2] for x in r])
        G[P + tag + "/eval"] = np.array([s.eval(c, v) for c in confs], np.float32)
        G[P + tag + "/ig_eval"] = np.array([s.ig_eval(c, v[1]) for c in confs], np.float32)
    r = [s.eval_deriv(c, V3, ig=1) for c in confs]
    G[P + "noncache/e"] = np.array([x[0] for x in r], np.float32)
    G[P + "noncache/change"] = np.stack([x[1] for x in r])
    G[P + "noncache/eval"] = np.array([s.eval(c, V3, ig=1) for c in confs], np.float32)
    G[P + "noncache/ig_eval"] = np.array([s.ig_eval(c, 1000.0, ig=1) for c in confs], np.float32)
    G[P + "within"] = np.array([s.within(c) for c in confs])
    if flex_text is None:
        fe = [s.final_energies(c) for c in confs]
        G[P + "final/e"] = np.array([x[0] for x in fe], np.float32)
        G[P + "final/intra"] = np.array([x[1] for x in fe], np.float32)
    mi = (25 + s.n_movable) // 3
    G[P + "max_iters"] = np.int32(mi)
    for tag, v in (("v1000", V3), ("v10", HUNT)):
        for iters in (1, 3, mi):
            r = [s.bfgs(c, v, max_iters=iters) for c in confs[:12]]
            G[P + f"bfgs/{tag}/{iters}/e"] = np.array([x[0] for x in r], np.float32)
            G[P + f"bfgs/{tag}/{iters}/conf"] = np.stack([x[1] for x in r])
            G[P + f"bfgs/{tag}/{iters}/grad"] = np.stack([x[2] for x in r])
    r = [s.bfgs(c, V3, ig=1, max_iters=mi) for c in confs[:12]]                     # quasi_newton on non_cache
    G[P + "bfgs_noncache/e"] = np.array([x[0] for x in r], np.float32)
    G[P + "bfgs_noncache/conf"] = np.stack([x[1] for x in r])
    for seed_, steps in mc:
        er, cr, xr = s.mc(seed_, steps, b, e, max_iters=mi, num_saved=20)
        G[P + f"mc/{seed_}_{steps}/e"], G[P + f"mc/{seed_}_{steps}/conf"], G[P + f"mc/{seed_}_{steps}/coords"] = er, cr, xr
    # short chains with two BFGS iterations from many seeds: what a device (other libm, other summation order) can
    # follow step for step (full-length minimisations amplify last-bit differences)
    for steps in (1, 3):
        rows = [s.mc(seed_, steps, b, e, max_iters=2, num_saved=20) for seed_ in range(100, 132)]
        G[P + f"mcshort/{steps}/n"] = np.array([len(r[0]) for r in rows], np.int32)
        G[P + f"mcshort/{steps}/e0"] = np.array([r[0][0] for r in rows], np.float32)
        G[P + f"mcshort/{steps}/conf0"] = np.stack([r[1][0] for r in rows])
    print(name, "atoms", s.n_atoms, "torsions", s.n_lig_tors, "pairs", s.n_lig_pairs, "box", n)


def main():
    if not ref.available():
        sys.exit("oracle/_ref cannot be built here (needs /root/reference)")
    rigid = open(RC.GSK3B).read()
    G = {}
    case(G, "adduct", rigid, RC.cys_adduct_ligand())
    case(G, "chain", rigid, RC.long_chain_ligand(), seed=1)
    # flexible side chain (the reference's flex_res_side_chain fixture) + a small ligand next to it
    case(G, "flex", rigid, RC.long_chain_ligand(n=8, origin=(-9.0, 12.0, 3.0)), seed=3,
         flex_text=open(RC.FLEX_RES).read(), mc=((1, 40),))
    # a box whose lattice hits szv_grid's 3 A cell boundaries exactly (degenerate candidate bricks)
    case(G, "aligned", rigid, RC.cys_adduct_ligand(), np.array([-7.5, 9.0, 0.7], np.float32),
         np.array([15.0, 16.0, 14.0], np.float32), seed=2, mc=((1, 40),))
    s = ref.Scene(rigid)
    rng = np.random.RandomState(9)
    r2 = np.concatenate([np.arange(0, 2049) / 32.0, rng.uniform(0, 64, 300)]).astype(np.float32)
    pairs = [(2, 2), (2, 13), (7, 13), (0, 5), (23, 12), (27, 17), (1, 12), (10, 4)]
    G["tables/r2"], G["tables/pairs"] = r2, np.array(pairs, np.int32)
    t = [s.table_eval(a, b, r2) for a, b in pairs]
    G["tables/fast"], G["tables/e"], G["tables/dor"] = (np.stack([x[k] for x in t]) for k in range(3))
    u, i, g = ref.random_stream(12345, 64)
    G["rng/uniform01"], G["rng/int_0_9"], G["rng/normal"] = u, i, g
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
