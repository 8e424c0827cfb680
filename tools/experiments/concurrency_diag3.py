"""Under concurrency: are the candidate lists (gather_pose_atoms) of a deviating call the quiet run's lists?"""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
capi.set_option("MI_GNINA_NO_LANES", "1")
L = capi.lib()
L.mi_debug_read_candidates.argtypes = [C.c_void_p] + [C.c_void_p] * 4
L.mi_debug_read_candidates.restype = C.c_int
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
victim_name, aggressor_name = sys.argv[1], sys.argv[2]
N = int(os.environ.get("DIAG_CALLS", "300"))
victim = capi.Scorer([victim_name]); victim.set_receptor(rec_xyz, rec_smt)
aggr = capi.Scorer([aggressor_name]); aggr.set_receptor(rec_xyz, rec_smt)


def cands(s):
    info = (C.c_int32 * 2)()
    capi.check(L.mi_debug_read_candidates(s.handle, info, None, None, None))
    ns, cap = info[0], info[1]
    cnt = np.zeros(ns, np.int32); ch = np.zeros(ns * cap, np.int32); rec = np.zeros(ns * cap * 8, np.float32)
    capi.check(L.mi_debug_read_candidates(s.handle, info, cnt.ctypes.data, ch.ctypes.data, rec.ctypes.data))
    ch = ch.reshape(ns, cap); rec = rec.reshape(ns, cap, 8)
    return cnt, [ch[i, :cnt[i]].copy() for i in range(ns)], [rec[i, :cnt[i]].copy() for i in range(ns)]


ref = {}
for p in range(len(poses)):
    r = victim.score_batch(poses[p:p + 1], lig_smt)
    ref[p] = ((float(r["pose"][0]), float(r["affinity"][0])), cands(victim))
    c = ref[p][1]
    print("pose", p, "counts", c[0], "channel lists sorted:", all((np.diff(x) >= 0).all() for x in c[1]))
stop = False


def aggressor():
    while not stop:
        aggr.score_batch(poses[:1], lig_smt)


th = threading.Thread(target=aggressor); th.start()
bad = 0; bad_lists = 0
for rep in range(N):
    p = rep % len(poses)
    r = victim.score_batch(poses[p:p + 1], lig_smt)
    sc = (float(r["pose"][0]), float(r["affinity"][0]))
    c = cands(victim)
    same = np.array_equal(c[0], ref[p][1][0]) and all(np.array_equal(a, b) for a, b in zip(c[1], ref[p][1][1])) and all(np.array_equal(a, b) for a, b in zip(c[2], ref[p][1][2]))
    if not same:
        bad_lists += 1
    if sc != ref[p][0]:
        bad += 1
        print("call", rep, "pose", p, "score diff", sc[0] - ref[p][0][0], sc[1] - ref[p][0][1], "candidate lists equal the quiet run's:", same)
stop = True; th.join()
print("deviating calls", bad, "calls with different candidate lists", bad_lists, "of", N)
