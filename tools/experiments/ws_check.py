"""conv3d_h2_ws_kernel (stationary weights + tile ring) against conv3d_h2_kernel: same bits?  how fast?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402

capi.init(0)
for name in (sys.argv[1:] or ["default2017", "dense"]):
    m = capi.Model(name)
    s = capi.Scorer([m])
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
    s.set_receptor(rec_xyz, rec_smt)
    for B in (() if os.environ.get("WS_NOBITS") else (1024, 67)):
        poses = synth.make_poses(rng, lx, B)
        capi.set_option("MI_GNINA_H2_WS", "0")
        ref = s.score_batch(poses, ls)
        for ring in (2, 3, 4, 5):
            capi.set_option("MI_GNINA_H2_WS", str(ring))
            got = s.score_batch(poses, ls)
            same = np.array_equal(got["pose"], ref["pose"]) and np.array_equal(got["affinity"], ref["affinity"])
            line = f"{name} B={B} ring {ring}: same bits {same}"
            if not same:
                line += f" max|dpose| {np.abs(got['pose'] - ref['pose']).max():.3g} max|daff| {np.abs(got['affinity'] - ref['affinity']).max():.3g} poses differing {int((got['pose'] != ref['pose']).sum())}"
            print(line, flush=True)
    poses = synth.make_poses(rng, lx, 1024)
    for ring, dbg in [(int(x.split(":")[0]), int(x.split(":")[1])) for x in os.environ.get("WS_RUNS", "0:0,2:0,3:0,4:0,5:0").split(",")]:
        capi.set_option("MI_GNINA_H2_WS", str(ring))
        capi.set_option("MI_GNINA_H2_DBG", str(dbg) if dbg else None)
        for _ in range(30):
            s.score_batch(poses, ls)
        t0 = time.perf_counter()
        for _ in range(30):
            s.score_batch(poses, ls)
        dt = (time.perf_counter() - t0) / 30
        s.enable_profile(True)
        s.score_batch(poses, ls)
        prof = s.profile()
        s.enable_profile(False)
        rows = prof if isinstance(prof, list) else prof.get("kernels", prof)
        conv1 = [r for r in rows if "conv3_s24" in r["kernel"]][:1]
        print(f"{name} ring {ring} dbg {dbg}: {1024 / dt:.0f} poses/s wall (host-output calls), first conv {conv1[0]['ms_total']:.3f} ms" if conv1 else f"{name} ring {ring}: {1024 / dt:.0f} poses/s", flush=True)
