#!/usr/bin/env python3
"""Goldens for the Overlap toy models (test/gnina/data/overlap.pt, overlap_smallr.pt: the models of the
reference's test/gnina/test_min.py).  Runs the reference TorchScript files on grids of the voxelizer oracle for the
reference's own test inputs (C.xyz / C1.xyz, CC.xyz / CC2.xyz) and a few random arrangements.
    python tests/golden/make_overlap_goldens.py      (build container: needs /root/reference)
Writes tests/golden/overlap_goldens.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cnn_ref, voxel  # noqa: E402

REF = "/root/reference/test/gnina/data/"
C = 2   # AliphaticCarbonXSHydrophobe


def main():
    rng = np.random.RandomState(0)
    cases = {"C_C1": (np.array([[0, 0, 0]], np.float32), np.array([[1, 1, 1]], np.float32)),
             "CC_CC2": (np.array([[0, 0, 0], [1.6, 0, 0]], np.float32), np.array([[1, -3.2, 1], [1, -1.6, 1]], np.float32)),
             "rand8": (rng.normal(0, 2, (8, 3)).astype(np.float32), rng.normal(0, 2, (5, 3)).astype(np.float32)),
             "far": (np.array([[0, 0, 0]], np.float32), np.array([[9, 9, 9]], np.float32))}   # no overlap -> 1e-20 branch
    out = {}
    for name in ("overlap", "overlap_smallr"):
        blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", name + ".mgw"))
        rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
        m = torch.jit.load(REF + name + ".pt", map_location="cpu")
        for cname, (rec, lig) in cases.items():
            rs, ls = np.full(len(rec), C, np.int32), np.full(len(lig), C, np.int32)
            grid, cen = voxel.voxelize_pose(rec, rs, lig, ls, rmap, lmap, None, blob.resolution, blob.dimension,
                                            blob.radius_scaling)
            with torch.no_grad():
                o, aff = m(torch.from_numpy(grid[None]))
            k = f"{name}/{cname}/"
            out[k + "rec"], out[k + "lig"] = rec, lig
            out[k + "pose"] = o[0, 1].numpy()                       # skip_softmax: pose = output[0, 1]
            out[k + "loss"] = (-torch.log(o[0, 1])).numpy()         # apply_logistic_loss
            out[k + "affinity"] = aff.numpy().reshape(())
            p2, a2, l2 = cnn_ref.scores(blob, grid[None])
            assert abs(float(p2[0]) - float(o[0, 1])) <= 1e-6 * max(1e-20, float(o[0, 1])) + 1e-30
            print(k, float(o[0, 1]), float(-torch.log(o[0, 1])))
    path = os.path.join(ROOT, "tests", "golden", "overlap_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
