"""two scorers of gnina's default ensemble on two threads, 150 B = 1 calls each (for a kernel trace: tools/experiments/r6_queues.sh)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi
capi.init(0)
for o in sys.argv[1:]:
    capi.set_option(*o.split("=", 1))
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
names = ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"]
rec_xyz, rec_smt, lig_smt, poses = (G[f"{names[0]}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
sc = []
for _ in range(int(os.environ.get("Q_THREADS", "2"))):
    s = capi.Scorer(names); s.set_receptor(rec_xyz, rec_smt)
    for _ in range(20): s.score_batch(poses[:1], lig_smt)
    sc.append(s)
N = 150
def loop(s):
    for k in range(N): s.score_batch(poses[k % 4:k % 4 + 1], lig_smt)
th = [threading.Thread(target=loop, args=(s,)) for s in sc]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
print(sys.argv[1:], f"{len(sc) * N / (time.perf_counter() - t0):.0f} poses/s from {len(sc)} threads")
