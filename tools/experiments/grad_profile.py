#!/usr/bin/env python3
"""Per-kernel breakdown (HIP events inside the engine) of a gradient call (score + d loss / d atoms: CNN refinement's
evaluation, torch_model.cpp:197-221) at the standard grid.   python tools/experiments/grad_profile.py [models...] [--batch N]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi, synth  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = 256
    for a in sys.argv[1:]:
        if a.startswith("--batch="):
            B = int(a.split("=")[1])
    capi.init(0)
    for name in (args or ["default2017", "crossdock_default2018", "dense"]):
        m = capi.Model(name)
        s = capi.Scorer([m])
        rng = np.random.RandomState(0)
        rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
        lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
        s.set_receptor(rec_xyz, rec_smt)
        poses = synth.make_poses(rng, lx, B)
        out = {}
        for label, f in (("forward", s.score_batch), ("forward_backward", s.score_grad)):
            f(poses, ls)
            t0 = time.perf_counter()
            for _ in range(3):
                f(poses, ls)
            out[label + "_poses_per_s"] = round(3 * B / (time.perf_counter() - t0), 1)
        print(json.dumps({"model": name, "batch": B, **out}))
        s.enable_profile(True)
        s.score_grad(poses, ls)
        prof = s.profile()
        s.enable_profile(False)
        rows = prof if isinstance(prof, list) else prof.get("kernels", prof)
        tot = 0.0
        for r in rows:
            tot += r["ms_total"]
            print("  %-44s %8.3f ms  %s" % (r["kernel"], r["ms_total"], ("%.1f TF" % (r["flops"] / (r["ms_total"] * 1e-3) / 1e12)) if r.get("flops") else ""))
        print("  sum of kernels %.3f ms" % tot)


if __name__ == "__main__":
    main()
