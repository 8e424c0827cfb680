#!/usr/bin/env python3
"""Gradient calls with the transposed convs on the split-fp16 kernel against the same calls on the fp32-MFMA kernels
(MI_GNINA_NO_H2_BWD=1 at run time): largest difference of the atom gradients relative to the largest gradient."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnina_amd import capi, synth  # noqa: E402

capi.init(0)
for name in sys.argv[1:] or ["default2017", "crossdock_default2018"]:
    m = capi.Model(name)
    s = capi.Scorer([m])
    rng = np.random.RandomState(0)
    rec_xyz, rec_smt = synth.make_receptor(rng, 2500, synth.mapped_types(m.chan_of_smt(False)))
    lx, ls = synth.make_ligand(rng, 32, synth.mapped_types(m.chan_of_smt(True)))
    s.set_receptor(rec_xyz, rec_smt)
    poses = synth.make_poses(rng, lx, 64)
    os.environ.pop("MI_GNINA_NO_H2_BWD", None)
    a = s.score_grad(poses, ls)
    os.environ["MI_GNINA_NO_H2_BWD"] = "1"
    b = s.score_grad(poses, ls)
    os.environ.pop("MI_GNINA_NO_H2_BWD", None)
    ga, gb = a["lig_grad"], b["lig_grad"]
    per_pose = np.abs(ga - gb).reshape(len(poses), -1).max(1) / np.abs(gb).reshape(len(poses), -1).max(1)
    print(json.dumps({"model": name, "max_rel_to_pose_max": float(per_pose.max()), "median": float(np.median(per_pose)),
                      "grad_max": float(np.abs(gb).max()), "equal_scores": bool(np.array_equal(a["pose"], b["pose"])),
                      "fallbacks": s.h2_fallbacks()}))
