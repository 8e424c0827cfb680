"""Soak: four host threads, each with its own scorer of gnina's default ensemble (lanes as the library decides), B = 1 ... 3
scoring calls and every fifth a gradient call, against the same scorers' single-thread outputs.  python tools/experiments/concurrency_soak.py [calls]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
pp = np.concatenate([poses, poses])


def call(s, rep):
    b, n = rep % 4, 1 + rep % 3
    if rep % 5 == 4:
        r = s.score_grad(pp[b:b + n], lig_smt)
        return np.concatenate([r["pose"], r["affinity"], r["lig_grad"].ravel()])
    r = s.score_batch(pp[b:b + n], lig_smt)
    return np.concatenate([r["pose"], r["affinity"]])


scorers = []
for _ in range(4):
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    scorers.append(s)
ref = [call(scorers[0], rep) for rep in range(60)]  # (the pattern repeats every 60 calls)
bad = [0] * 4


def loop(k):
    for rep in range(N):
        if not np.array_equal(call(scorers[k], rep), ref[rep % 60]):
            bad[k] += 1


for nt in (1, 2, 4):
    th = [threading.Thread(target=loop, args=(k,)) for k in range(nt)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(f"{nt} thread(s) x {N} calls: deviating calls per thread {bad[:nt]}")
    bad = [0] * 4
