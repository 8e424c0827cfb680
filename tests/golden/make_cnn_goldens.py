#!/usr/bin/env python3
"""Generate CNN-score goldens from the reference's OWN TorchScript models.

Run in the build container (needs /root/reference and torch):
    python tests/golden/make_cnn_goldens.py
For each committed weight blob, builds the seeded synthetic complex of gnina_amd/synth.py,
voxelizes 4 poses with the (golden-pinned) voxelizer oracle, runs
torch.jit.load('/root/reference/gninasrc/lib/models/<name>.pt') on the grids -- i.e. the
reference's module.forward (gninasrc/lib/torch_model.cpp:185) -- applies the reference's
post-processing (torch_model.cpp:188-195) and stores inputs (atoms) + outputs.
Writes tests/golden/cnn_goldens.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import synth  # noqa: E402
from oracle import cnn_ref, voxel  # noqa: E402

MODELS = {  # blob name -> reference file stem
    "default2017": "default2017",
    "crossdock_default2018": "crossdock_default2018",
    "dense": "dense",
    "dense_1_3": "dense_1.3",
    "dense_1_3_PT_KD_3": "dense_1.3_PT_KD_3",
    "crossdock_default2018_KD_4": "crossdock_default2018_KD_4",
}
REF = "/root/reference/gninasrc/lib/models/"
N_POSES = 4


def main():
    out = {}
    for name, stem in MODELS.items():
        blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", name + ".mgw"))
        rmap = voxel.typer_parse(blob.recmap_text())
        lmap = voxel.typer_parse(blob.ligmap_text())
        rec_xyz, rec_smt, lig_smt, poses = synth.make_complex(
            1, synth.mapped_types(rmap[0]), synth.mapped_types(lmap[0]), n_rec=2500, n_lig=32, n_poses=N_POSES)
        grids = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap,
                                              None, blob.resolution, blob.dimension, blob.radius_scaling)[0]
                          for b in range(N_POSES)])
        m = torch.jit.load(REF + stem + ".pt", map_location="cpu")
        with torch.no_grad():
            logp, aff = m(torch.from_numpy(grids))
            pose = torch.softmax(logp, 1)[:, 1]
            loss = torch.nn.functional.cross_entropy(logp, torch.ones(N_POSES, dtype=torch.long), reduction="none")
            m64 = torch.jit.load(REF + stem + ".pt", map_location="cpu").double()
            logp64, aff64 = m64(torch.from_numpy(grids).double())
            pose64 = torch.softmax(logp64, 1)[:, 1]
        out[name + "/rec_xyz"] = rec_xyz
        out[name + "/rec_smt"] = rec_smt
        out[name + "/lig_smt"] = lig_smt
        out[name + "/poses"] = poses
        out[name + "/logp"] = logp.numpy()
        out[name + "/pose"] = pose.numpy()
        out[name + "/affinity"] = aff.numpy()
        out[name + "/loss"] = loss.numpy()
        out[name + "/pose64"] = pose64.numpy()
        out[name + "/affinity64"] = aff64.numpy()
        out[name + "/grid_sum"] = grids.reshape(N_POSES, -1).sum(1, dtype=np.float64)
        out[name + "/grid_nnz"] = (grids.reshape(N_POSES, -1) != 0).sum(1)
        print(name, "pose", pose.numpy(), "aff", aff.numpy(), "nnz frac", (grids != 0).mean())
    path = os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
