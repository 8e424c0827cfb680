"""Which LAYER of the victim deviates first when another scorer runs on a second host thread: after every concurrent call of
the victim its activation buffers (mi_debug_read_activation) are compared with the ones a quiet run left for the same pose."""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402
from oracle import cnn_ref  # noqa: E402  (buffer list of the model only)

capi.init(0)
capi.set_option("MI_GNINA_NO_LANES", "1")
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
victim_name, aggressor_name = sys.argv[1], sys.argv[2]
N = int(os.environ.get("DIAG_CALLS", "120"))

victim = capi.Scorer([victim_name])
victim.set_receptor(rec_xyz, rec_smt)
aggr = capi.Scorer([aggressor_name])
aggr.set_receptor(rec_xyz, rec_smt)
blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", victim_name + ".mgw"))
print("ops:", [" ".join(t[:8]) for t in blob.ops])
bufs = []
victim.score_batch(poses[:1], lig_smt)
for b in sorted(blob.bufs):
    try:
        a, split = victim.read_activation(b, 1)
        bufs.append(b)
    except Exception:
        pass
print("readable buffers", bufs)
ref = {}
for p in range(len(poses)):
    r = victim.score_batch(poses[p:p + 1], lig_smt)
    ref[p] = ({b: victim.read_activation(b, 1)[0] for b in bufs}, (float(r["pose"][0]), float(r["affinity"][0])))
stop = False


def aggressor():
    while not stop:
        aggr.score_batch(poses[:1], lig_smt)


th = threading.Thread(target=aggressor)
th.start()
bad = 0
for rep in range(N):
    p = rep % len(poses)
    r = victim.score_batch(poses[p:p + 1], lig_smt)
    sc = (float(r["pose"][0]), float(r["affinity"][0]))
    acts = {b: victim.read_activation(b, 1)[0] for b in bufs}
    first = None
    if os.environ.get("DIAG_DETAIL") and bufs:
        b0 = bufs[0]
        d = np.abs(acts[b0] - ref[p][0][b0])
        if d.max() > 0:
            prev_p = (rep - 1) % len(poses)
            for i in np.argwhere(d > 0)[:16]:
                i = tuple(i)
                print("   call", rep, "pose", p, "buf", b0, "idx", i[1:], "got", acts[b0][i], "want", ref[p][0][b0][i],
                      "previous pose's value there", ref[prev_p][0][b0][i], "other poses", [float(ref[q][0][b0][i]) for q in range(len(poses))])
    for b in bufs:
        d = np.abs(acts[b] - ref[p][0][b])
        if d.max() > 0:
            idx = np.argwhere(d > 0)
            ch = sorted(set(int(i[-1]) for i in idx))
            vox = sorted(set((int(i[1]), int(i[2]), int(i[3])) for i in idx))
            first = (b, len(idx), float(d.max()), "channels", ch[:12], len(ch), "voxels", vox[:6], len(vox))
            break
    if first or sc != ref[p][1]:
        bad += 1
        if bad <= 12:
            print("call", rep, "score diff", sc[0] - ref[p][1][0], sc[1] - ref[p][1][1], "first deviating buffer:", first)
stop = True
th.join()
print("deviating calls:", bad, "of", N)
