"""What does a second hardware queue do to the voxelizer?  (round 6)

The victim is the voxelizer ALONE: one scorer runs gather_pose_atoms + voxelize_tiles for one pose again and again
(mi_debug_vox_stress: no network behind it, no per-device lock) and every iteration's pooled grid is compared ON THE DEVICE
with the grid the same scorer produced while nothing else ran.  The aggressor is a second scorer on a second host thread
scoring B = 1 poses in a loop.  Switches select what the aggressor's kernels do (their timing switches: results wrong, which
does not matter here) and what the victim does between iterations.

    python tools/experiments/vox_stress.py [--victim crossdock_default2018_KD_4] [--aggressor dense_1_3] [--iters 3000]
        [--flags N]       victim: 2 = gather once, 4 = poison the grid before every iteration, 8 = trap ring (needs the
                          -DMI_VOX_TRAP library: MI_GNINA_LIB=gnina_amd/lib/variants/libmi_gnina_trap.so)
        [--opt NAME=V ...]  mi_gnina_set_option before the aggressor starts (e.g. MI_GNINA_D16_DBG=12: no LDS-DMA in d16)
        [--aggr-path f32]   aggressor on the fp32-MFMA program
        [--no-aggressor]
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--victim", default="crossdock_default2018_KD_4")
ap.add_argument("--aggressor", default="dense_1_3")
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--aggr-path", default="")
ap.add_argument("--aggr-batch", type=int, default=1)
ap.add_argument("--no-aggressor", action="store_true")
ap.add_argument("--label", default="")
a = ap.parse_args()

capi.init(0)
capi.set_option("MI_GNINA_NO_LANES", "1")
capi.set_option("MI_GNINA_NO_CALL_LOCK", "1")
L = capi.lib()
L.mi_debug_vox_stress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
L.mi_debug_vox_stress.restype = C.c_int
G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
lig_smt = np.ascontiguousarray(lig_smt, np.int32)
pose0 = np.ascontiguousarray(poses[0], np.float32)

victim = capi.Scorer([a.victim])
victim.set_receptor(rec_xyz, rec_smt)
LOGCAP = 4096
log = np.zeros((LOGCAP, 4), np.int32)
trap = np.zeros((1024, 16), np.uint32)


def stress(iters, flags):
    capi.check(L.mi_debug_vox_stress(victim.handle, pose0.ctypes.data, lig_smt.ctypes.data, len(lig_smt), iters, flags,
                                     log.ctypes.data, LOGCAP, trap.ctypes.data))


stress(1, 1)  # the reference grid, nothing else running
stress(200, a.flags & ~8)
quiet = int(log[0, 0])

stop = False
calls = [0]
if not a.no_aggressor:
    for o in a.opt:
        k, v = o.split("=", 1)
        capi.set_option(k, v)
    aggr = capi.Scorer([a.aggressor])
    aggr.set_receptor(rec_xyz, rec_smt)
    if a.aggr_path == "f32":
        aggr.set_precision(2)
    ab = np.ascontiguousarray(np.repeat(poses[:1], a.aggr_batch, axis=0))

    def aggressor():
        while not stop:
            aggr.score_batch(ab, lig_smt)
            calls[0] += 1

    th = threading.Thread(target=aggressor)
    th.start()
    time.sleep(0.2)
t0 = time.time()
stress(a.iters, a.flags)
dt = time.time() - t0
stop = True
if not a.no_aggressor:
    th.join()
nd = int(log[0, 0])
rows = log[1:min(nd, LOGCAP - 1) + 1]
its = sorted(set(int(r[0]) for r in rows))
print(f"[{a.label or ' '.join(sys.argv[1:])}] quiet: {quiet} differing dwords of 200 iterations; next to the aggressor: {nd} differing dwords in "
      f"{len(its)} of {a.iters} iterations ({dt:.2f} s, aggressor calls {calls[0]})")
S = 24
for it in its[:12]:
    r = rows[rows[:, 0] == it]
    desc = []
    for _, idx, got, want in r[:16]:
        idx = int(idx)
        octet, rem = divmod(idx, S * S * S * 8)
        cell, dw = divmod(rem, 8)
        x, y, z = cell // (S * S), (cell // S) % S, cell % S
        g16 = np.array([got & 0xffff, (got >> 16) & 0xffff], np.uint16).view(np.float16)
        w16 = np.array([want & 0xffff, (want >> 16) & 0xffff], np.uint16).view(np.float16)
        desc.append(f"oct{octet} cell({x},{y},{z}) dw{dw}({'h' if dw < 4 else 'l'} ch{octet * 8 + 2 * (dw & 3)},+1) got {g16[0]:.4g},{g16[1]:.4g} want {w16[0]:.4g},{w16[1]:.4g}")
    print(f"  iteration {it}: {len(r)} dwords:", "; ".join(desc))
if a.flags & 8:
    nt = int(trap[0, 0])
    print(f"  trap records: {nt}")
    kinds = {1: "channel below current", 2: "scalar != vector record", 3: "LDS canary", 5: "window index", 8: "windows at end"}
    for r in trap[1:min(nt, 1023) + 1][:40]:
        hw = int(r[9])
        print(f"    {kinds.get(int(r[0]), int(r[0]))}: wg {int(r[1])} tile/where {int(r[2]):#x} detail {[hex(int(x)) for x in r[3:9]]} "
              f"wave {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7} xcc {int(r[10]) & 15} t {int(r[11]) | int(r[12]) << 32}")
