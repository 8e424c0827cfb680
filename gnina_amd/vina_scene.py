"""Shared builder of the synthetic Vina scene used by the oracle tests and the GPU parity tests
(2,500-atom receptor with a 4 A pocket, 32-atom ligand with 6 torsions,
box = ligand bounding box + 4 A padding at 0.375 A, main.cpp:622-634)."""
import numpy as np

from gnina_amd import synth


def build(seed=0, n_rec=2500, n_atoms=32, n_tors=6, pad=4.0):
    rng = np.random.RandomState(seed)
    rec_types = np.array([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 23], dtype=np.int32)
    rec_xyz, rec_smt = synth.make_receptor(rng, n_rec, rec_types)
    lig = synth.make_ligand_tree(rng, n_atoms, n_tors)
    # centre the ligand's reference conformation in the pocket
    shift = -lig["coords0"].mean(0)
    lig["coords0"] = (lig["coords0"] + shift).astype(np.float32)
    lig["conf0"][:3] += shift
    lo, hi = lig["coords0"].min(0) - pad, lig["coords0"].max(0) + pad
    center, size = (lo + hi) / 2, hi - lo
    return dict(rec_xyz=rec_xyz, rec_smt=rec_smt, lig=lig, center=center.astype(np.float32),
                size=size.astype(np.float32), rng=rng)
