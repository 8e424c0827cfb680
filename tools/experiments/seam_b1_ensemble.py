"""B = 1 calls of gnina's default ensemble through the C ABI: median wall time per call from one thread, and throughput from
1 / 2 / 4 host threads with a scorer each (fresh_copy() per worker thread, main.cpp:1436-1438)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
for o in sys.argv[1:]:
    capi.set_option(*o.split("=", 1))
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))


def make():
    s = capi.Scorer(names)
    s.set_receptor(rec_xyz, rec_smt)
    for _ in range(30):
        s.score_batch(poses[:1], lig_smt)
    return s


s0 = make()
t = []
for k in range(300):
    t0 = time.perf_counter()
    s0.score_batch(poses[k % 4:k % 4 + 1], lig_smt)
    t.append(time.perf_counter() - t0)
print(f"{sys.argv[1:]} one thread: median {np.median(t) * 1e6:.0f} us per call, min {np.min(t) * 1e6:.0f}")
for nt in (1, 2, 4):
    scorers = [s0] + [make() for _ in range(nt - 1)]
    N = 400

    def loop(s):
        for k in range(N):
            s.score_batch(poses[k % 4:k % 4 + 1], lig_smt)

    th = [threading.Thread(target=loop, args=(s,)) for s in scorers]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print(f"  {nt} thread(s): {nt * N / dt:.0f} poses/s")
