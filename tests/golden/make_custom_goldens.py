"""Goldens for the generic TorchScript path (`--cnn_model file.pt` of an architecture that is not a shipped family): the three
architectures of tests/custom_models.py are scripted, saved with gnina's metadata, converted by
gnina_amd/tools/extract_weights.py (its graph walk) and run BY TORCH ITSELF on the oracle's grids of the seeded golden
complex.  Writes tests/golden/custom_<kind>.mgw (what the engine loads) and tests/golden/custom_goldens.npz (what it must
score).  Run from the repo root: python tests/golden/make_custom_goldens.py"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gnina_amd.tools import extract_weights  # noqa: E402
from oracle import cnn_ref, voxel  # noqa: E402
from tests import custom_models  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "cnn_goldens.npz"))
base = "dense_1_3"  # the seeded complex these goldens share (default 28-channel maps)
rec_xyz, rec_smt, lig_smt, poses = (G[f"{base}/{k}"] for k in ("rec_xyz", "rec_smt", "lig_smt", "poses"))
out = {}
for kind in ("stack", "minidense", "postact"):
    with tempfile.TemporaryDirectory() as d:
        pt = os.path.join(d, f"custom_{kind}.pt")
        m = custom_models.save_scripted(kind, pt)
        data, name = extract_weights.convert(pt)
    open(os.path.join(ROOT, "tests", "golden", f"custom_{kind}.mgw"), "wb").write(data)
    blob = cnn_ref.Blob(data)
    rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
    grids = np.stack([voxel.voxelize_pose(rec_xyz, rec_smt, p, lig_smt, rmap, lmap)[0] for p in poses])
    with torch.no_grad():
        logp, aff = m(torch.from_numpy(grids))
        lp2, aff2 = cnn_ref.module_output(blob, grids)
    assert float((logp - lp2).abs().max()) < 1e-5 and float((aff - aff2).abs().max()) < 1e-5
    out[f"{kind}/pose"] = torch.softmax(logp, 1)[:, 1].numpy()   # torch_model.cpp:189
    out[f"{kind}/affinity"] = aff.numpy()
    print(kind, len(data), "bytes", out[f"{kind}/pose"], out[f"{kind}/affinity"])
np.savez(os.path.join(ROOT, "tests", "golden", "custom_goldens.npz"), **out)
