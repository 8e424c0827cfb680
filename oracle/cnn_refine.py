"""CPU restatement of CNN-in-the-loop optimisation: non_cache_cnn (gninasrc/lib/non_cache_cnn.cpp:33-54,
79-169) as the igrid of quasi_newton, and refine_structure (gninasrc/main/main.cpp:131-171) on it.
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Built from the pinned pieces: the voxelizer oracle
(voxel_ref.c), the CNN oracle with autograd (cnn_ref.py) and the torsion-tree / BFGS oracle (vina_ref.c)."""
import numpy as np

from . import cnn_ref, vina, voxel

MAX_FL = np.float32(3.402823466e+38)


class NonCacheCnn:
    """dl_scorer + the two penalty boxes.  blobs: list of cnn_ref.Blob (the ensemble)."""

    def __init__(self, blobs, rec_xyz, rec_smt, lig, search_box=None, cnn_dimension=23.5, mix_emp_force=False,
                 mix_emp_energy=False, empirical_weight=1.0, tables=None, v=1000.0, per_atom_forces=False):
        self.blobs, self.rec_xyz, self.rec_smt, self.lig = blobs, rec_xyz, rec_smt, lig
        self.mix_emp_force, self.mix_emp_energy, self.weight = mix_emp_force, mix_emp_energy, empirical_weight
        self.tables, self.v = tables, v       # precalculate_linear tables (oracle.vina.Tables) for the empirical term
        self.smt = lig.arr["smt"]
        self.search_box = search_box          # (begin[3], end[3]) or None
        self.cnn_dimension = cnn_dimension
        self.cnn_center = None                # set by adjust_center
        self.slope = 10.0
        self.evals = 0
        self.user_grid = None                 # (vina.GridDims, data [(n+1)^3]) of --user_grid, or None
        self.per_atom_forces = per_atom_forces   # False = model::add_minus_forces' indexing (see eval_deriv)
        # gnina's combined model (tree.h / model.h): atoms = [movable side chains | ligand | inflex]; a plain ligand
        # has lig_begin = 0, lig_end = n_movable = n_atoms.  rec_xyz / rec_smt are then DLScorer::setReceptor's rows:
        # [movable side-chain atoms | inflex | rigid].
        self.n_movable = lig.n_movable
        self.lig_begin = int(lig.c.lig_begin) if lig.c.lig_end > lig.c.lig_begin else 0
        self.lig_end = int(lig.c.lig_end) if lig.c.lig_end > lig.c.lig_begin else lig.n_atoms

    def adjust_center(self, conf):
        """DLScorer::set_center_from_model (dl_scorer.cpp:197-217): fp32 mean of the heavy movable atoms"""
        coords, _, _ = vina.set_conf(self.lig, conf)
        c = np.zeros(3, dtype=np.float32)
        cnt = 0
        for i in range(self.n_movable):
            if self.smt[i] > 1:
                c = (c + coords[i]).astype(np.float32)
                cnt += 1
        self.cnn_center = (c / np.float32(cnt)).astype(np.float32)

    def _cnn(self, coords, deriv):
        """-> (loss, gradient indexed like the model's atoms).  Combined model (flexible residues): the movable
        side-chain atoms are the first rows of the scorer's receptor and are refreshed from the model on every call
        (DLScorer::setReceptor, dl_scorer.cpp:150-193), the ligand is atoms [lig_begin, lig_end)
        (setLigand, :71-87); both gradients return by movable-atom index (cnn_torch_scorer.cpp:208-228)."""
        loss_sum, grad = 0.0, np.zeros((len(self.smt), 3), dtype=np.float64)
        lb, le = self.lig_begin, self.lig_end
        rec_xyz = self.rec_xyz
        if lb > 0:
            rec_xyz = self.rec_xyz.copy()
            rec_xyz[:lb] = coords[:lb]
        lig_xyz, lig_smt = coords[lb:le], self.smt[lb:le]
        for blob in self.blobs:
            rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
            # (the blob's own grid: a dynamic-pool model re-gridded by the test -- blob.resolution / .dimension overwritten,
            # like gnina's metadata override for dense_1.3 at 0.25 A -- is voxelized at that grid)
            grid, cen = voxel.voxelize_pose(rec_xyz, self.rec_smt, lig_xyz, lig_smt, rmap, lmap, None, blob.resolution,
                                            blob.dimension, blob.radius_scaling)
            if deriv:
                loss, gg = cnn_ref.loss_and_grid_gradient(blob, grid[None])
                ch, rad = voxel.type_atoms(lig_smt, lmap[0])
                ch = np.where(ch >= 0, ch + rmap[1], -1)
                grad[lb:le] += voxel.grid_backward(cen, lig_xyz, ch, rad, rmap[1] + lmap[1], gg[0].numpy(), blob.resolution,
                                                   blob.dimension, blob.radius_scaling)
                if lb > 0:
                    chr_, radr = voxel.type_atoms(self.rec_smt[:lb], rmap[0])
                    grad[:lb] += voxel.grid_backward(cen, rec_xyz[:lb], chr_, radr, rmap[1] + lmap[1], gg[0].numpy(),
                                                     blob.resolution, blob.dimension, blob.radius_scaling)
                loss_sum += float(loss[0])
            else:
                loss_sum += float(cnn_ref.scores(blob, grid[None])[2][0])
        n = len(self.blobs)
        return loss_sum / n, (grad / n).astype(np.float32)

    def _bounds(self, c):
        """check_bounds_deriv on gd, then on cnn_gd (non_cache.cpp:102-123) -> (penalty, force)"""
        pen, f = np.float32(0), np.zeros(3, dtype=np.float32)
        boxes = []
        if self.search_box is not None:
            boxes.append(self.search_box)
        if self.cnn_center is not None:
            h = np.float32(self.cnn_dimension / 2.0)
            boxes.append((self.cnn_center - h, self.cnn_center + h))
        for lo, hi in boxes:
            dist = np.float32(0)
            for k in range(3):
                if c[k] < lo[k]:
                    f[k] += -self.slope
                    dist += abs(c[k] - np.float32(lo[k]))
                elif c[k] > hi[k]:
                    f[k] += self.slope
                    dist += abs(c[k] - np.float32(hi[k]))
            pen += dist * np.float32(self.slope)
        return pen, f

    def _empirical(self, i, c):
        """non_cache_cnn.cpp:113-137: sum over receptor atoms within the cutoff of the atom clamped to the search
        box -> (energy, derivative, search-box penalty direction * slope), before curl"""
        adj, oob = np.array(c, dtype=np.float32), np.zeros(3, dtype=np.float32)
        if self.search_box is not None:
            lo, hi = self.search_box
            for k in range(3):
                if c[k] < lo[k]:
                    adj[k], oob[k] = lo[k], -self.slope
                elif c[k] > hi[k]:
                    adj[k], oob[k] = hi[k], self.slope
        r = adj[None, :] - self.rec_xyz
        r2 = (r * r).sum(1, dtype=np.float32)
        e, d = np.float32(0), np.zeros(3, dtype=np.float32)
        for j in np.nonzero(r2 < 64.0)[0]:
            ej, dor = self.tables.eval_deriv(int(self.smt[i]), int(self.rec_smt[j]), float(r2[j]))
            e += np.float32(ej)
            d += np.float32(dor) * r[j]
        return e, d, oob

    def eval_deriv(self, conf):
        self.evals += 1
        coords, _, _ = vina.set_conf(self.lig, conf)
        loss, grad = self._cnn(coords, True)
        e = np.float32(loss)
        forces = np.zeros_like(coords)
        w = np.float32(self.weight)
        rank = 0                               # model::add_minus_forces' counter j (model.cu:247-259)
        for i in range(self.n_movable):
            if self.smt[i] <= 1:
                continue                       # hydrogens: minus_forces = 0
            pen, f = self._bounds(coords[i])
            # the scorer's gradient has one entry per movable atom, hydrogens included (cnn_torch_scorer.cpp:208-228);
            # add_minus_forces gives the k-th NON-HYDROGEN atom entry k -- pinned by gnina's own code running on
            # HipCNNScorer (tests/cpp/test_cnn_dropin.cpp).  per_atom_forces: each atom its own entry instead.
            forces[i] = grad[i if self.per_atom_forces else rank] + f
            rank += 1
            emp_e = np.float32(0)
            uge, ugd = np.float32(0), np.zeros(3, dtype=np.float32)
            if self.user_grid is not None:      # non_cache_cnn.cpp:141-151: this_e / deriv, curled on their own
                uge, ugd = vina.grid_evaluate(self.user_grid[0], self.user_grid[1], coords[i], self.slope, 1000.0)
                uge, ugd = np.float32(uge), ugd.astype(np.float32)
                ce, cd = uge, ugd.copy()
                if ce > 0 and self.v < 0.1 * MAX_FL:
                    tmp = np.float32(self.v / (self.v + ce))
                    ce, cd = ce * tmp, cd * tmp * tmp
                forces[i] = forces[i] + cd
                e += ce
            if self.mix_emp_force:
                emp_e, emp_d, oob = self._empirical(i, coords[i])
                emp_e, emp_d = emp_e + uge, emp_d + ugd     # :146-149
                if emp_e > 0 and self.v < 0.1 * MAX_FL:          # curl (curl.h:29-42)
                    tmp = np.float32(self.v / (self.v + emp_e))
                    emp_e, emp_d = emp_e * tmp, emp_d * tmp * tmp
                forces[i] = (forces[i] + w * (emp_d + oob)) / (np.float32(1) + w)
            e += pen
            if self.mix_emp_energy:
                e += w * emp_e
        if self.mix_emp_energy:
            e /= (np.float32(1) + w)
        change, _ = vina.forces_to_change(self.lig, conf, forces)
        return float(e), change

    def eval(self, conf):
        coords, _, _ = vina.set_conf(self.lig, conf)
        loss, _ = self._cnn(coords, False)
        e = np.float32(loss)
        for i in range(self.n_movable):
            if self.smt[i] > 1:
                e += self._bounds(coords[i])[0]
        return float(e)

    def within(self, conf, margin=1e-4):
        """non_cache_cnn::within = gd_within(cnn_gd) || non_cache::within (non_cache_cnn.cpp:74-76)"""
        coords, _, _ = vina.set_conf(self.lig, conf)
        heavy = coords[:self.n_movable][self.smt[:self.n_movable] > 1]
        h = self.cnn_dimension / 2.0
        in_cnn = bool(((heavy >= self.cnn_center - h - margin) & (heavy <= self.cnn_center + h + margin)).all())
        in_box = True
        if self.search_box is not None:
            lo, hi = self.search_box
            in_box = bool(((heavy >= np.asarray(lo) - margin) & (heavy <= np.asarray(hi) + margin)).all())
        return in_cnn or in_box


def refine_structure(nc, conf, max_iters):
    """main.cpp:131-171 -> (energy, conf, tries)"""
    nc.adjust_center(conf)
    slope, e, tries = 10.0, 0.0, 0
    conf = np.array(conf, dtype=np.float32, copy=True)
    for _ in range(5):
        nc.slope = slope
        e, conf, _, _ = vina.bfgs_callback(nc.lig, conf, nc.eval_deriv, max_iters)
        tries += 1
        if nc.within(conf):
            break
        slope *= 10
    if not nc.within(conf):
        e = float(MAX_FL)
    return e, conf, tries


# ---------------------------------------------------------------------------------------------------------------
# --cnn_scoring all: monte_carlo::operator() (monte_carlo.cpp:99-148) with non_cache_cnn as BOTH igrids
# (parallel_mc.cpp:156-159).  Small cases only: every evaluation is a PyTorch CPU forward (+ backward).
# ---------------------------------------------------------------------------------------------------------------
import ctypes as _C

_libm = _C.CDLL("libm.so.6")
for _n in ("logf", "cosf", "sqrtf", "expf"):
    getattr(_libm, _n).restype = _C.c_float
    getattr(_libm, _n).argtypes = [_C.c_float]


class Mt19937:
    """boost::mt19937 under the restated Boost distributions (oracle/ref_shims/boost/random.hpp, random.cpp:27-75)"""

    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(int(seed) & 0xffffffff)

    def u32(self):
        return int(self.bg.random_raw())

    def fl(self, a, b):
        a, b = np.float32(a), np.float32(b)
        while True:
            r = np.float32(np.float32(np.float32(self.u32()) / np.float32(4294967296.0)) * np.float32(b - a) + a)
            if r < b:
                return r

    def irange(self, a, b):
        rng, brange = b - a, 0xffffffff
        if rng == 0:
            return a
        bucket = brange // (rng + 1)
        if brange % (rng + 1) == rng:
            bucket += 1
        while True:
            q = self.u32() // bucket
            if q <= rng:
                return q + a

    def normal(self):
        r1, r2 = self.fl(0, 1), self.fl(0, 1)
        return np.float32(np.float32(_libm.sqrtf(np.float32(np.float32(-2.0) * np.float32(_libm.logf(np.float32(1) - r2))))) *
                          np.float32(_libm.cosf(np.float32(np.float32(2.0) * np.float32(3.14159265358979323846) * r1))))

    def inside_sphere(self):
        while True:
            v = np.array([self.fl(-1, 1), self.fl(-1, 1), self.fl(-1, 1)], dtype=np.float32)
            if np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) < 1:
                return v


def mc_cnnall(nc, seed, n_steps, corner1, corner2, max_iters, num_saved=50, temperature=1.2, amplitude=2.0,
              min_rmsd=1.0, hunt_v=10.0, auth_v=1000.0):
    """-> list of (energy, conf, heavy coords) sorted by energy, and the number of eval_deriv calls"""
    lig, smt = nc.lig, nc.smt
    nt = lig.n_tors
    heavy = [i for i in range(len(smt)) if smt[i] > 1]
    rng = Mt19937(seed)
    PI = np.float32(3.14159265358979323846)
    tmp = np.zeros(7 + nt, dtype=np.float32)
    for k in range(3):
        tmp[k] = rng.fl(corner1[k], corner2[k])
    while True:
        q = np.array([rng.normal() for _ in range(4)], dtype=np.float32)
        mx = np.abs(q).max()
        nrm = np.float32(0)
        if mx != 0:
            inv = np.float32(1.0 / float(mx))
            s = np.float32(0)
            for k in range(4):
                s = np.float32(s + np.float32(q[k] * inv) * np.float32(q[k] * inv))
            nrm = np.float32(mx * np.float32(_libm.sqrtf(s)))
        if nrm > np.float32(1.1920928955078125e-07):
            break
    tmp[3:7] = q / nrm
    for t in range(nt):
        tmp[7 + t] = rng.fl(-PI, PI)
    mconf = np.zeros(7 + nt, dtype=np.float32)
    mconf[3] = 1
    nc.cnn_center = None                     # a fresh non_cache_cnn: no cube until the first adjust_center
    tmp_e, best_e, out, evals = np.float32(0), MAX_FL, [], 0
    last = {}

    def fx(c):
        last["c"] = np.array(c, dtype=np.float32, copy=True)
        return nc.eval_deriv(c)

    def minimise(x, v):
        nonlocal evals
        nc.v = v
        last["c"] = np.array(x, dtype=np.float32, copy=True)
        e, xo, _, ev = vina.bfgs_callback(lig, x, fx, max_iters)
        evals += ev
        return xo, last["c"]

    def update_energy(m):
        nc.adjust_center(m)
        nc.v = auth_v
        return np.float32(nc.eval(m))

    for step in range(n_steps):
        cand = tmp.copy()
        which = rng.irange(0, 2 + nt - 1)
        if which == 0:
            cand[:3] = (cand[:3] + np.float32(amplitude) * rng.inside_sphere()).astype(np.float32)
        elif which == 1:
            co, _, _ = vina.set_conf(lig, mconf)
            acc = np.float32(0)
            for i in heavy:
                d = (co[i] - mconf[:3]).astype(np.float32)
                acc = np.float32(acc + np.float32(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
            gr = np.float32(_libm.sqrtf(np.float32(acc / np.float32(len(heavy)))))
            if gr > np.float32(1.1920928955078125e-07):
                rot = (np.float32(np.float32(amplitude) / gr) * rng.inside_sphere()).astype(np.float32)
                cand = vina.conf_increment(cand, np.concatenate([np.zeros(3, np.float32), rot]), 1.0, 0)
        else:
            cand[7 + which - 2] = rng.fl(-PI, PI)
        cand, mconf = minimise(cand, hunt_v)
        cand_e = update_energy(mconf)
        accept = step == 0 or cand_e < tmp_e
        if not accept:
            accept = rng.fl(0, 1) < np.float32(_libm.expf(np.float32((tmp_e - cand_e) / np.float32(temperature))))
        if not accept:
            continue
        tmp, tmp_e, mconf = cand.copy(), cand_e, cand.copy()
        if tmp_e < best_e or len(out) < num_saved:
            tmp, mconf = minimise(tmp, auth_v)
            tmp_e = update_energy(mconf)
            mconf = tmp.copy()
            co, _, _ = vina.set_conf(lig, tmp)
            hc = co[heavy].copy()
            closest, closest_rmsd = len(out), MAX_FL
            for o, (_, _, x) in enumerate(out):
                r = np.float32(np.sqrt(np.float32(((hc - x) ** 2).sum(dtype=np.float32) / np.float32(len(heavy)))))
                if o == 0 or r < closest_rmsd:
                    closest, closest_rmsd = o, r
            if closest < len(out) and closest_rmsd < min_rmsd:
                if tmp_e < out[closest][0]:
                    out[closest] = (tmp_e, tmp.copy(), hc)
            elif len(out) < num_saved:
                out.append((tmp_e, tmp.copy(), hc))
            elif out and tmp_e < out[-1][0]:
                out[-1] = (tmp_e, tmp.copy(), hc)
            out.sort(key=lambda t: t[0])
            if tmp_e < best_e:
                best_e = tmp_e
    return out, evals
