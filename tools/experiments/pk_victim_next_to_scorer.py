"""The packed-fp32 microbenchmark's victims (tools/microbench/pk_f32_next_to_mfma.hip built as libpk_victim.so) next to a REAL
aggressor: a Dense scorer of libmi_gnina.so scoring B = 1 poses on a second host thread.  Which victim instruction forms go
wrong next to the kernels that make voxelize_tiles go wrong?"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

capi.init(0)
capi.set_option("MI_GNINA_NO_LANES", "1")
capi.set_option("MI_GNINA_NO_CALL_LOCK", "1")
V = C.CDLL(os.path.join(ROOT, "tools", "microbench", "libpk_victim.so"))
V.pk_victim_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
rec_xyz, rec_smt, lig_smt, poses = (G[f"dense_1_3/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
args = [a for a in sys.argv[1:] if "=" not in a]
for o in sys.argv[1:]:
    if "=" in o:  # NAME=value: a library switch (the aggressor's timing switches)
        capi.set_option(*o.split("=", 1))
MODES = [int(m) for m in os.environ.get("PK_MODES", "0,1,2,3,4,5,6,7,8,9").split(",")]
aggr = capi.Scorer([args[0] if args else "dense_1_3"])
print("aggressor:", args[0] if args else "dense_1_3", [o for o in sys.argv[1:] if "=" in o], flush=True)
aggr.set_receptor(rec_xyz, rec_smt)
stop = False
calls = [0]


def aggressor():
    while not stop:
        aggr.score_batch(poses[:1], lig_smt)
        calls[0] += 1


names = ["pk_add g - s[a]", "pk_add g - v[a]", "pk_mul d * d", "pk_add x + y.lo", "the voxelizer's chain (compiler)", "pk_add g - s[a], s_load in flight",
         "pk_add g - s[a], s_load waited for", "the chain in one asm, hipcc's s_nop 0", "the chain, s_nop 3 behind every instruction",
         "the chain, s_nop 7 behind every instruction"]
bad = np.zeros(2, np.uint64)
for quiet in (True, False):
    if not quiet:
        th = threading.Thread(target=aggressor)
        th.start()
        time.sleep(0.2)
    for mode, nm in enumerate(names):
        if mode not in MODES:
            continue
        t0 = time.time()
        V.pk_victim_run(mode, 60, 216, 3000, bad.ctypes.data)
        print(f"{'quiet' if quiet else 'next to the scorer'}: {nm:40s} wrong results lanes 0-31: {int(bad[0])}, lanes 32-63: {int(bad[1])}  ({time.time() - t0:.2f} s, "
              f"aggressor calls so far {calls[0]})", flush=True)
stop = True
th.join()
