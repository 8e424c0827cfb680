"""Deterministic synthetic receptor/ligand/pose generator (bench + parity inputs).

Mirrors the reference's random-molecule test fixture (test/gnina/test_utils.cpp:12-44 `make_mol`:
uniform coordinates, uniform smina types) with the concrete C2 definition of SURVEY §8(d) /
BASELINE.md §4: receptor atoms uniform in [-20,20]^3 A with an empty 4 A pocket at the origin,
ligand heavy atoms ~ N(0, 2.5^2 I), types uniform over the smina types the model's map accepts,
poses = uniform random rotation (normalised 4-Gaussian quaternion, quaternion.cu:81-94 style)
about the ligand centroid + translation uniform in [-2,2]^3.
"""
import numpy as np

NUM_SMINA_TYPES = 28


def mapped_types(chan_of_smt):
    return np.nonzero(np.asarray(chan_of_smt) >= 0)[0].astype(np.int32)


def make_receptor(rng, n_atoms, types, half_box=20.0, pocket=4.0):
    xyz = np.empty((n_atoms, 3), dtype=np.float32)
    i = 0
    while i < n_atoms:
        p = rng.uniform(-half_box, half_box, size=(n_atoms, 3))
        p = p[np.linalg.norm(p, axis=1) >= pocket]
        k = min(len(p), n_atoms - i)
        xyz[i:i + k] = p[:k]
        i += k
    smt = rng.choice(types, size=n_atoms).astype(np.int32)
    return xyz, smt


def make_ligand(rng, n_atoms, types, sigma=2.5):
    xyz = rng.normal(0.0, sigma, size=(n_atoms, 3)).astype(np.float32)
    smt = rng.choice(types, size=n_atoms).astype(np.int32)
    return xyz, smt


def quat_to_matrix(q):
    a, b, c, d = q
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]], dtype=np.float64)


def make_poses(rng, lig_xyz, n_poses, max_trans=2.0):
    """Rigid poses of one ligand: returns float32 [n_poses, L, 3]."""
    cen = lig_xyz.astype(np.float64).mean(0)
    out = np.empty((n_poses, len(lig_xyz), 3), dtype=np.float32)
    for b in range(n_poses):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        R = quat_to_matrix(q)
        t = rng.uniform(-max_trans, max_trans, size=3)
        out[b] = ((lig_xyz - cen) @ R.T + cen + t).astype(np.float32)
    return out


def make_complex(seed, rec_types, lig_types, n_rec=2500, n_lig=32, n_poses=1024):
    """The C2 workload: (rec_xyz, rec_smt, lig_smt, poses[n_poses, n_lig, 3])."""
    rng = np.random.RandomState(seed)
    rec_xyz, rec_smt = make_receptor(rng, n_rec, rec_types)
    lig_xyz, lig_smt = make_ligand(rng, n_lig, lig_types)
    poses = make_poses(rng, lig_xyz, n_poses)
    return rec_xyz, rec_smt, lig_smt, poses
