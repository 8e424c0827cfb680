"""Do two scorers on two host threads (lanes off: every call serial on its own stream) disturb each other?  gnina scores from
several threads, each with its own DLScorer copy.  Per-call outputs against the same scorer's single-threaded outputs."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
capi.set_option("MI_GNINA_NO_LANES", "1")
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))


def loop(s, reps, out):
    for rep in range(reps):
        b = rep % len(poses)
        r = s.score_batch(poses[b:b + 1], lig_smt)
        out.append((float(r["pose"][0]), float(r["affinity"][0])))


PAIRS = (["dense_1_3", "crossdock_default2018_KD_4"], ["dense_1_3", "dense_1_3_PT_KD_3"], ["default2017", "crossdock_default2018_KD_4"],
         ["crossdock_default2018", "crossdock_default2018_KD_4"], ["dense_1_3", "default2017"])
if len(sys.argv) > 1:
    PAIRS = [a.split(",") for a in sys.argv[1:]]
N = int(os.environ.get("DIAG_CALLS", "60"))
for pair in PAIRS:
    scorers = []
    for n in pair:
        s = capi.Scorer([n])
        s.set_receptor(rec_xyz, rec_smt)
        scorers.append(s)
    refs = []
    for s in scorers:
        o = []
        loop(s, N, o)
        refs.append(np.array(o))
    outs = [[] for _ in scorers]
    th = [threading.Thread(target=loop, args=(s, N, o)) for s, o in zip(scorers, outs)]
    for t in th: t.start()
    for t in th: t.join()
    for n, o, r in zip(pair, outs, refs):
        d = np.abs(np.array(o) - r)
        print(pair, n, "calls that deviate under concurrency:", int((d.max(axis=1) > 0).sum()), "of", N, "max |d|", float(d.max()))
