#!/usr/bin/env python3
"""Per-call latency of the B = 1 path (what gnina's per-pose DLScorer::score calls see) and small batches,
host pointers, synchronous -- the reference's own usage pattern (one pose per call)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnina_amd import capi, synth  # noqa: E402


def main():
    capi.init(0)
    rng = np.random.RandomState(0)
    m0 = capi.Model("crossdock_default2018")
    rt, lt = synth.mapped_types(m0.chan_of_smt(False)), synth.mapped_types(m0.chan_of_smt(True))
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, rt)
    lx, ls = synth.make_ligand(rng, 32, lt)
    for label, models in (("default2017", ["default2017"]),
                          ("default ensemble", ["dense_1_3", "dense_1_3_PT_KD_3", "crossdock_default2018_KD_4"])):
        s = capi.Scorer(models)
        s.set_receptor(rec_xyz, rec_smt)
        row = {"models": label}
        for B in (1, 9, 50):
            poses = synth.make_poses(rng, lx, B)
            for grad in (False, True):
                f = (lambda: s.score_grad(poses, ls)) if grad else (lambda: s.score_batch(poses, ls))
                for _ in range(5):
                    f()
                n = 50
                t0 = time.perf_counter()
                for _ in range(n):
                    f()
                row[f"B{B}_{'fwd_bwd' if grad else 'fwd'}_us"] = round((time.perf_counter() - t0) / n * 1e6, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
