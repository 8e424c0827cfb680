"""GPU probe (run on the MI355X box): how close to the reference's bits are the Vina kernels, default vs strict order?
Prints and writes gpurun_out/strict_probe.json: sincos vs libm, cache grids, eval_deriv, BFGS 1 / 3 / full, MC chains.
    python tools/experiments/strict_parity_probe.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
G = np.load(os.path.join(ROOT, "tests", "golden", "vina_goldens.npz"))
out = {}

# 1. sincos_ref vs the host's sinf / cosf
libm = C.CDLL("libm.so.6")
libm.sinf.restype = C.c_float
libm.sinf.argtypes = [C.c_float]
libm.cosf.restype = C.c_float
libm.cosf.argtypes = [C.c_float]
rng = np.random.RandomState(0)
x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-1e-3, 1e-3, 20000), rng.uniform(-100, 100, 20000),
                    [0.0, -0.0, np.pi / 4, np.pi / 2, -np.pi / 2, 2.0 ** -12, 2.0 ** -13]]).astype(np.float32)
sn, cs = np.empty_like(x), np.empty_like(x)
f32p = C.POINTER(C.c_float)
capi.check(capi.lib().mi_debug_sincos(x.ctypes.data_as(f32p), len(x), sn.ctypes.data_as(f32p), cs.ctypes.data_as(f32p)))
s0 = np.array([libm.sinf(float(v)) for v in x], np.float32)
c0 = np.array([libm.cosf(float(v)) for v in x], np.float32)
out["sincos"] = {"n": int(len(x)), "sin_mismatch": int((sn.view(np.uint32) != s0.view(np.uint32)).sum()),
                 "cos_mismatch": int((cs.view(np.uint32) != c0.view(np.uint32)).sum())}
print("sincos", out["sincos"], flush=True)
libm.expf.restype = libm.logf.restype = C.c_float
libm.expf.argtypes = libm.logf.argtypes = [C.c_float]
xe = np.concatenate([rng.uniform(-104, 0, 100000), rng.uniform(0, 88, 20000), -np.exp(rng.uniform(-20, 4.6, 50000))]).astype(np.float32)
_, _, ex, _ = capi.device_libm(xe)
e0 = np.array([libm.expf(float(v)) for v in xe], np.float32)
xl = np.concatenate([rng.uniform(0, 1, 100000), np.exp(rng.uniform(-17, 3, 50000)), [1.0, 2.0 ** -24]]).astype(np.float32)
xl = xl[xl > 1e-30]
_, _, _, lg = capi.device_libm(xl)
l0 = np.array([libm.logf(float(v)) for v in xl], np.float32)
out["explog"] = {"exp_mismatch": int((ex.view(np.uint32) != e0.view(np.uint32)).sum()), "n_exp": int(len(xe)),
                 "log_mismatch": int((lg.view(np.uint32) != l0.view(np.uint32)).sum()), "n_log": int(len(xl))}
print("explog", out["explog"], flush=True)

V3, HUNT = (1000.0, 1000.0, 1000.0), (10.0, 10.0, 10.0)


def biteq(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))


for name in ("adduct", "chain"):
    P = name + "/"
    R = out[name] = {}
    lig = capi.read_pdbqt_ligand(bytes(G[P + "lig_text"]).decode(), is_text=True)
    v = capi.Vina()
    v.set_receptor(G[P + "rec_xyz"], G[P + "rec_smt"])
    v.build_cache(list(G[P + "begin"]), list(G[P + "end"]), [int(k) for k in G[P + "n"]], [int(t) for t in G[P + "types"]], 1e3)
    v.set_ligand(lig)
    idx = G[P + "grid_idx"]
    R["grid_biteq"] = [biteq(v.cache_grid(int(t))[idx[:, 2], idx[:, 1], idx[:, 0]], G[P + "grid_val"][k])
                       for k, t in enumerate(G[P + "types"])]
    confs = G[P + "confs"]
    mi = int(G[P + "max_iters"])
    seeds = np.arange(100, 132, dtype=np.uint64)
    for mode in ("default", "strict"):
        v.set_strict_order(mode == "strict")
        M = R[mode] = {}
        for tag, cap in (("v1000", V3), ("v10", HUNT)):
            e, ch, co = v.eval_batch(confs, cap, deriv=True, want_coords=True)
            M[tag] = {"coords": biteq(co, G[P + tag + "/coords"]),
                      "e_eq": int(sum(biteq(e[b], G[P + tag + "/e"][b]) for b in range(len(confs)))),
                      "change_eq": int(sum(biteq(ch[b], G[P + tag + "/change"][b]) for b in range(len(confs)))),
                      "n": int(len(confs)),
                      "max_rel_e": float(np.max(np.abs(e - G[P + tag + "/e"]) / np.maximum(1, np.abs(G[P + tag + "/e"])))),
                      "eval_eq": int(sum(biteq(a, b) for a, b in zip(v.eval_batch(confs, cap, deriv=False)[0], G[P + tag + "/eval"]))),
                      "ig_eval_eq": int(sum(biteq(a, b) for a, b in zip(v.eval_batch(confs, cap, grid_only=True)[0], G[P + tag + "/ig_eval"])))}
            for iters in (1, 3, mi):
                e, cf, g, ev = v.bfgs_batch(confs[:12], cap, max_iters=iters)
                e0, c0 = G[P + f"bfgs/{tag}/{iters}/e"], G[P + f"bfgs/{tag}/{iters}/conf"]
                g0 = G[P + f"bfgs/{tag}/{iters}/grad"]
                M[f"bfgs/{tag}/{iters}"] = {
                    "biteq": int(sum(biteq(e[b], e0[b]) and biteq(cf[b], c0[b]) for b in range(12))),
                    "grad_biteq": int(sum(biteq(g[b], g0[b]) for b in range(12))),
                    "close": int(sum(abs(e[b] - e0[b]) <= 1e-3 * max(1.0, abs(e0[b])) and np.abs(cf[b] - c0[b]).max() < 1e-2
                                     for b in range(12)))}
        e, ch, _ = v.eval_batch(confs, V3, deriv=True, direct=True)
        M["noncache"] = {"e_eq": int(sum(biteq(e[b], G[P + "noncache/e"][b]) for b in range(len(confs)))),
                         "change_eq": int(sum(biteq(ch[b], G[P + "noncache/change"][b]) for b in range(len(confs))))}
        for steps in (1, 3):
            n, e, cf, xyz, ev = v.mc_batch(seeds, list(G[P + "begin"]), list(G[P + "end"]), capi.McParams.default(steps, 2, 20))
            e0, c0, n0 = G[P + f"mcshort/{steps}/e0"], G[P + f"mcshort/{steps}/conf0"], G[P + f"mcshort/{steps}/n"]
            M[f"mcshort/{steps}"] = {
                "biteq": int(sum(biteq(e[b, 0], e0[b]) and biteq(cf[b, 0], c0[b]) and n[b] == n0[b] for b in range(32))),
                "close": int(sum(abs(e[b, 0] - e0[b]) <= 1e-3 * max(1.0, abs(e0[b])) and np.abs(cf[b, 0] - c0[b]).max() < 1e-2
                                 for b in range(32)))}
        for key in [k for k in G.files if k.startswith(P + "mc/") and k.endswith("/e")]:
            seed_, steps = (int(t) for t in key.split("/")[2].split("_"))
            n, e, cf, xyz, ev = v.mc_batch(np.array([seed_], np.uint64), list(G[P + "begin"]), list(G[P + "end"]),
                                           capi.McParams.default(steps, mi, 20))
            e0, c0 = G[key], G[key[:-2] + "/conf"]
            k = min(int(n[0]), len(e0))
            M[f"mc/{seed_}_{steps}"] = {"n": int(n[0]), "n_ref": int(len(e0)),
                                        "biteq": bool(int(n[0]) == len(e0) and biteq(e[0, :k], e0[:k]) and biteq(cf[0, :k], c0[:k])),
                                        "best": float(e[0, 0]), "best_ref": float(e0[0])}
        print(name, mode, json.dumps(M), flush=True)
    v.set_strict_order(False)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "strict_probe.json"), "w"), indent=1)
