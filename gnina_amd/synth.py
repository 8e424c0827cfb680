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


# ----------------------------------------------------------------------------------------------
# Flexible ligand with a Vina torsion tree (chain topology, T rotatable bonds)
# ----------------------------------------------------------------------------------------------
def make_ligand_tree(rng, n_atoms=32, n_tors=6, types=None, bond=1.5):
    """Random chain-like molecule cut into n_tors+1 rigid fragments, laid out the way gnina's PDBQT
    parser builds `model.ligands[0]` (gninasrc/lib/parse_pdbqt.cpp:343-380, tree.h:206-233):
      * nodes in DFS pre-order, node 0 = rigid root, node k>0 owns torsion k-1;
      * the first atom of a branch (on the rotation axis) is stored with the PARENT node;
      * node origin = that axis atom, axis = unit(origin - axis_begin);
      * atoms are stored node by node, local coordinates relative to the node origin;
      * interacting pairs follow model::initialize_pairs (model.cpp:682-703): i<j, both heavy,
        relative distance VARIABLE (not in one node, not an axis atom vs. the node it turns),
        and not within 3 bonds.
    Returns a dict of numpy arrays (int32 / float32)."""
    if types is None:
        types = np.array([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18], dtype=np.int32)
    # 1. self-avoiding backbone with short side branches
    xyz = [np.zeros(3)]
    bonds = []
    attach = [0]
    while len(xyz) < n_atoms:
        src = attach[-1] if rng.rand() < 0.8 else attach[rng.randint(len(attach))]
        for _ in range(200):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            p = xyz[src] + bond * d
            if all(np.linalg.norm(p - q) > 1.3 for k, q in enumerate(xyz) if k != src):
                break
        xyz.append(p)
        bonds.append((src, len(xyz) - 1))
        attach.append(len(xyz) - 1)
    xyz = np.array(xyz)
    n = n_atoms
    adj = [[] for _ in range(n)]
    for a, b in bonds:
        adj[a].append(b)
        adj[b].append(a)
    # 2. choose rotatable bonds: both sides must keep >= 2 atoms
    def side(a, b):  # atoms on b's side when bond a-b is cut
        seen, stack = {b}, [b]
        while stack:
            u = stack.pop()
            for w in adj[u]:
                if w not in seen and not (u == b and w == a):
                    seen.add(w)
                    stack.append(w)
        return seen
    cand = [(a, b) for a, b in bonds if 2 <= len(side(a, b)) <= n - 2]
    rng.shuffle(cand)
    rot = []
    for a, b in cand:
        if len(rot) == n_tors:
            break
        if all(a not in r and b not in r for r in rot):  # keep rotors non-adjacent
            rot.append((a, b))
    rotset = {frozenset(r) for r in rot}
    # 3. fragments = components without the rotatable bonds; tree rooted at atom 0's fragment
    frag = -np.ones(n, dtype=int)
    nf = 0
    for s in range(n):
        if frag[s] >= 0:
            continue
        frag[s] = nf
        stack = [s]
        while stack:
            u = stack.pop()
            for w in adj[u]:
                if frag[w] < 0 and frozenset((u, w)) not in rotset:
                    frag[w] = nf
                    stack.append(w)
        nf += 1
    # DFS over fragments
    nodes = []  # dict(frag, parent, axis_begin, axis_end)
    def visit(f, parent, ab, ae):
        k = len(nodes)
        nodes.append(dict(frag=f, parent=parent, axis_begin=ab, axis_end=ae))
        for a, b in rot:
            for u, w in ((a, b), (b, a)):
                if frag[u] == f and frag[w] != f and all(nd["frag"] != frag[w] for nd in nodes):
                    visit(frag[w], k, u, w)
    visit(frag[0], -1, -1, -1)
    # 4. node atom lists: fragment atoms, except that a branch's axis_end atom lives in its parent node
    owner = np.array([next(k for k, nd in enumerate(nodes) if nd["frag"] == frag[i]) for i in range(n)])
    for k, nd in enumerate(nodes):
        if k > 0:
            owner[nd["axis_end"]] = nd["parent"]
    order = np.concatenate([np.nonzero(owner == k)[0] for k in range(len(nodes))]).astype(int)
    newidx = np.empty(n, dtype=int)
    newidx[order] = np.arange(n)
    abeg, aend = [], []
    pos = 0
    for k in range(len(nodes)):
        cnt = int((owner == k).sum())
        abeg.append(pos)
        aend.append(pos + cnt)
        pos += cnt
    coords = xyz[order].astype(np.float32)
    node_of = owner[order]
    origin = np.zeros((len(nodes), 3), dtype=np.float32)
    rel_origin = np.zeros((len(nodes), 3), dtype=np.float32)
    rel_axis = np.zeros((len(nodes), 3), dtype=np.float32)
    origin[0] = coords[abeg[0]]
    for k, nd in enumerate(nodes):
        if k == 0:
            continue
        o = coords[newidx[nd["axis_end"]]]
        ab = coords[newidx[nd["axis_begin"]]]
        origin[k] = o
        d = (o - ab).astype(np.float32)
        rel_axis[k] = d / np.float32(np.linalg.norm(d))
    for k, nd in enumerate(nodes):
        if k > 0:
            rel_origin[k] = origin[k] - origin[nd["parent"]]
    local = (coords - origin[node_of]).astype(np.float32)
    smt = rng.choice(types, size=n).astype(np.int32)
    # 5. interacting pairs
    nadj = [[newidx[w] for w in adj[order[i]]] for i in range(n)]
    def within3(i):
        seen, frontier = {i}, {i}
        for _ in range(3):
            frontier = {w for u in frontier for w in nadj[u]} - seen
            seen |= frontier
        return seen
    fixed = np.zeros((n, n), dtype=bool)
    for k, nd in enumerate(nodes):
        idx = np.arange(abeg[k], aend[k])
        fixed[np.ix_(idx, idx)] = True
        if k > 0:
            for ax in (newidx[nd["axis_begin"]], newidx[nd["axis_end"]]):
                fixed[ax, idx] = True
                fixed[idx, ax] = True
    pairs = []
    for i in range(n):
        near = within3(i)
        for j in range(i + 1, n):
            if not fixed[i, j] and j not in near and smt[i] > 1 and smt[j] > 1:
                pairs.append((i, j))
    return dict(
        smt=smt, local_xyz=local, coords0=coords,
        parent=np.array([nd["parent"] for nd in nodes], dtype=np.int32),
        abeg=np.array(abeg, dtype=np.int32), aend=np.array(aend, dtype=np.int32),
        rel_origin=rel_origin, rel_axis=rel_axis,
        pairs=np.array(pairs, dtype=np.int32).reshape(-1, 2),
        conf0=np.concatenate([origin[0], [1, 0, 0, 0], np.zeros(len(nodes) - 1)]).astype(np.float32),
        n_tors=len(nodes) - 1)


def random_conf(rng, lig, center, spread=2.0):
    """A random conformation: position near `center`, random unit quaternion, torsions U(-pi, pi)."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([np.asarray(center) + rng.uniform(-spread, spread, 3), q,
                           rng.uniform(-np.pi, np.pi, lig["n_tors"])]).astype(np.float32)
