#!/usr/bin/env python3
"""Goldens for the 0.25 A / 96^3 configuration (BASELINE config 5): the reference's own dense_1.3.pt
(dynamic global max pool) run on 96^3 grids of the seeded synthetic complex.

Run in the build container (needs /root/reference and torch):
    python tests/golden/make_cnn_goldens_96.py
Atoms come from the dense_1_3 entry of cnn_goldens.npz; grids from the golden-pinned voxelizer oracle at
resolution 0.25 / dimension 23.75.  Writes tests/golden/cnn_goldens_96.npz (outputs only).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cnn_ref, voxel  # noqa: E402

REF = "/root/reference/gninasrc/lib/models/"
RES, DIM, N_POSES = 0.25, 23.75, 2


def main():
    G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
    out = {}
    for name, stem in (("dense_1_3", "dense_1.3"), ("dense_1_3_PT_KD_3", "dense_1.3_PT_KD_3")):
        blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", name + ".mgw"))
        rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
        rec_xyz, rec_smt, lig_smt, poses = (G[f"{name}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
        grids = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, None, RES, DIM,
                                              blob.radius_scaling)[0] for b in range(N_POSES)])
        assert grids.shape[-1] == 96
        m = torch.jit.load(REF + stem + ".pt", map_location="cpu")
        with torch.no_grad():
            logp, aff = m(torch.from_numpy(grids))
            pose = torch.softmax(logp, 1)[:, 1]
            loss = torch.nn.functional.cross_entropy(logp, torch.ones(N_POSES, dtype=torch.long), reduction="none")
        out[name + "/pose"] = pose.numpy()
        out[name + "/affinity"] = aff.numpy()
        out[name + "/loss"] = loss.numpy()
        out[name + "/grid_sum"] = grids.reshape(N_POSES, -1).sum(1, dtype=np.float64)
        out[name + "/grid_nnz"] = (grids.reshape(N_POSES, -1) != 0).sum(1)
        # the restated network must agree with the reference module on the same grids
        p2, a2, l2 = cnn_ref.scores(blob, grids)
        assert np.abs(p2.numpy() - pose.numpy()).max() < 1e-5 and np.abs(a2.numpy() - aff.numpy()).max() < 1e-4
        print(name, "pose", pose.numpy(), "aff", aff.numpy(), "loss", loss.numpy())
    out["resolution"], out["dimension"] = np.float32(RES), np.float32(DIM)
    path = os.path.join(ROOT, "tests", "golden", "cnn_goldens_96.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
