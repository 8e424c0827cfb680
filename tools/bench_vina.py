#!/usr/bin/env python3
"""Vina-path throughput on the GPU (report evaluations/s and chains in flight, not a
roofline fraction): model::eval_deriv evaluations/s and BFGS evaluations/s for B conformations of
the C3 synthetic complex (2,500-atom receptor, 32-atom / 6-torsion ligand), next to the CPU oracle
on one host core."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402
from gnina_amd import vina_scene  # noqa: E402


def main():
    capi.init(0)
    sc = vina_scene.build(0)
    lig = sc["lig"]
    from oracle import vina as V   # grid dims helper + CPU baseline only
    gd = V.setup_grid_dims(sc["center"], sc["size"])
    types = sorted(set(int(t) for t in lig["smt"] if t > 1))
    vina = capi.Vina()
    vina.set_receptor(sc["rec_xyz"], sc["rec_smt"])
    t0 = time.perf_counter()
    vina.build_cache(list(gd.begin), list(gd.end), list(gd.n), types, 1e3)
    t_cache = time.perf_counter() - t0
    vina.set_ligand(lig)
    rng = np.random.RandomState(1)
    res = {"cache_build_s": round(t_cache, 4), "grid_points": int(np.prod([n + 1 for n in gd.n])),
           "n_types": len(types), "eval": [], "bfgs": []}
    v = (10.0, 10.0, 10.0)
    for B in (64, 1024, 8192, 65536):
        confs = np.stack([synth.random_conf(rng, lig, sc["center"], 1.0) for _ in range(min(B, 2048))])
        confs = np.tile(confs, (B // len(confs) + 1, 1))[:B]
        vina.eval_batch(confs, v)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            vina.eval_batch(confs, v)
        dt = (time.perf_counter() - t0) / reps
        res["eval"].append({"B": B, "ms": round(1e3 * dt, 3), "evals_per_s": round(B / dt)})
    for B in (64, 1024, 8192):
        confs = np.stack([synth.random_conf(rng, lig, sc["center"], 1.0) for _ in range(min(B, 2048))])
        confs = np.tile(confs, (B // len(confs) + 1, 1))[:B]
        vina.bfgs_batch(confs, v)
        t0 = time.perf_counter()
        e, cf, g, ev = vina.bfgs_batch(confs, v)
        dt = time.perf_counter() - t0
        res["bfgs"].append({"B": B, "ms": round(1e3 * dt, 2), "evals": int(ev.sum()), "evals_per_s": round(ev.sum() / dt),
                            "us_per_eval_per_chain": round(1e6 * dt / ev.max(), 2)})
    c64 = np.stack([synth.random_conf(rng, lig, sc["center"], 1.0) for _ in range(64)])
    res["eval_latency_us_B64"] = {name: round(vina.eval_latency_us(c64, mode), 2) for name, mode in
                                  (("coords_only", 3), ("receptor_grid_only", 2), ("pairs_only", 4),
                                   ("energy", 0), ("energy_and_gradient", 1))}
    # CPU oracle, one core
    T = V.Tables()
    grids = {t: V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], t) for t in types[:1]}
    t0 = time.perf_counter()
    V.cache_populate(T, gd, sc["rec_xyz"], sc["rec_smt"], types[0])
    res["cpu_cache_build_s_per_type"] = round(time.perf_counter() - t0, 3)
    grids = {t: (grids[types[0]] if t == types[0] else vina.cache_grid(t)) for t in types}
    S = V.Scene(T, gd, grids, V.LigandHandle(lig))
    confs = [synth.random_conf(rng, lig, sc["center"], 1.0) for _ in range(16)]
    t0 = time.perf_counter()
    n_ev = 0
    for c in confs:
        n_ev += S.bfgs(c, v)[3]
    dt = time.perf_counter() - t0
    res["cpu_oracle_bfgs_evals_per_s_1core"] = round(n_ev / dt)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
