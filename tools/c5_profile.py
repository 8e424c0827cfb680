#!/usr/bin/env python3
"""Per-kernel breakdown (HIP events inside the engine) of BASELINE config C5's network on one GPU:
dense_1_3 at 0.25 A (96^3), B poses, fp32 or bf16.   python tools/c5_profile.py [--bf16] [--batch 256] [--grad]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnina_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--grad", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="dense_1_3")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    capi.init(0)
    m = capi.Model(a.model, resolution=0.25, dimension=23.75)
    s = capi.Scorer([m])
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
    s.set_receptor(rec_xyz, rec_smt)
    poses = synth.make_poses(rng, lx, a.batch)
    s.set_precision(a.bf16)
    f = s.score_grad if a.grad else s.score_batch
    f(poses, ls)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        f(poses, ls)
    dt = (time.perf_counter() - t0) / a.reps
    s.enable_profile(True)
    f(poses, ls)
    prof = s.profile()
    s.enable_profile(False)
    rows = prof if isinstance(prof, list) else prof.get("kernels", prof)
    print(json.dumps({"poses_per_s": round(a.batch / dt, 1), "ms_per_call": round(dt * 1e3, 3), "bf16": a.bf16,
                      "grad": a.grad}))
    tot = 0.0
    for r in rows:
        ms = r["ms_total"]
        tot += ms
        print("  %-44s %8.3f ms  %s" % (r["kernel"], ms,
                                         ("%.1f TF" % (r["flops"] / (ms * 1e-3) / 1e12)) if r.get("flops") else ""))
    print("  sum of kernels %.3f ms" % tot)


if __name__ == "__main__":
    main()
