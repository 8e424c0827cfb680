#!/usr/bin/env python3
"""CNN-score goldens for ALL of the reference's built-in models (cnn_torch_scorer.cpp:24-64), from the reference's
OWN TorchScript files.  Same recipe as make_cnn_goldens.py; the atoms are stored once per distinct (recmap, ligmap)
pair, per model only the outputs of the reference's module.forward + post-processing on 4 poses.

Run in the build container (needs /root/reference, torch and the extracted blobs: python -m gnina_amd.build):
    python tests/golden/make_cnn_goldens_all.py
Writes tests/golden/cnn_goldens_all.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd import synth  # noqa: E402
from oracle import cnn_ref, voxel  # noqa: E402

REF = "/root/reference/gninasrc/lib/models/"
N_POSES = 4


def main():
    out, sig_of, grids_of = {}, {}, {}
    names = []
    for f in sorted(os.listdir(REF)):
        if not f.endswith(".pt"):
            continue
        stem = f[:-3]
        name = stem.replace(".", "_")
        blob = cnn_ref.Blob(os.path.join(ROOT, "gnina_amd", "weights", name + ".mgw"))
        key = (blob.recmap_text(), blob.ligmap_text(), blob.resolution, blob.dimension, blob.radius_scaling)
        if key not in sig_of:
            sig = f"sig{len(sig_of)}"
            sig_of[key] = sig
            rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
            rec_xyz, rec_smt, lig_smt, poses = synth.make_complex(
                7, synth.mapped_types(rmap[0]), synth.mapped_types(lmap[0]), n_rec=2500, n_lig=32, n_poses=N_POSES)
            grids_of[sig] = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, poses[b], lig_smt, rmap, lmap, None,
                                                          blob.resolution, blob.dimension, blob.radius_scaling)[0]
                                      for b in range(N_POSES)])
            out[sig + "/rec_xyz"], out[sig + "/rec_smt"] = rec_xyz, rec_smt
            out[sig + "/lig_smt"], out[sig + "/poses"] = lig_smt, poses
        sig = sig_of[key]
        m = torch.jit.load(REF + f, map_location="cpu")
        with torch.no_grad():
            logp, aff = m(torch.from_numpy(grids_of[sig]))
            pose = torch.softmax(logp, 1)[:, 1]
            loss = torch.nn.functional.cross_entropy(logp, torch.ones(N_POSES, dtype=torch.long), reduction="none")
        out[name + "/pose"], out[name + "/affinity"], out[name + "/loss"] = pose.numpy(), aff.numpy(), loss.numpy()
        out[name + "/sig"] = np.array(int(sig[3:]), np.int32)
        names.append(name)
        print(name, sig, pose.numpy().round(4), aff.numpy().round(3))
    out["names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "cnn_goldens_all.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(names), "models,", len(sig_of), "atom sets")


if __name__ == "__main__":
    main()
