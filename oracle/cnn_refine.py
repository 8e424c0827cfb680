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
                 mix_emp_energy=False, empirical_weight=1.0, tables=None, v=1000.0):
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

    def adjust_center(self, conf):
        """DLScorer::set_center_from_model (dl_scorer.cpp:197-217): fp32 mean of the heavy movable atoms"""
        coords, _, _ = vina.set_conf(self.lig, conf)
        c = np.zeros(3, dtype=np.float32)
        cnt = 0
        for i in range(len(self.smt)):
            if self.smt[i] > 1:
                c = (c + coords[i]).astype(np.float32)
                cnt += 1
        self.cnn_center = (c / np.float32(cnt)).astype(np.float32)

    def _cnn(self, coords, deriv):
        loss_sum, grad = 0.0, np.zeros((len(self.smt), 3), dtype=np.float64)
        for blob in self.blobs:
            rmap, lmap = voxel.typer_parse(blob.recmap_text()), voxel.typer_parse(blob.ligmap_text())
            grid, cen = voxel.voxelize_pose(self.rec_xyz, self.rec_smt, coords, self.smt, rmap, lmap)
            if deriv:
                loss, gg = cnn_ref.loss_and_grid_gradient(blob, grid[None])
                ch, rad = voxel.type_atoms(self.smt, lmap[0])
                ch = np.where(ch >= 0, ch + rmap[1], -1)
                grad += voxel.grid_backward(cen, coords, ch, rad, rmap[1] + lmap[1], gg[0].numpy(), blob.resolution,
                                            blob.dimension, blob.radius_scaling)
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
        for i in range(len(self.smt)):
            if self.smt[i] <= 1:
                continue                       # hydrogens: minus_forces = 0
            pen, f = self._bounds(coords[i])
            forces[i] = grad[i] + f
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
        for i in range(len(self.smt)):
            if self.smt[i] > 1:
                e += self._bounds(coords[i])[0]
        return float(e)

    def within(self, conf, margin=1e-4):
        """non_cache_cnn::within = gd_within(cnn_gd) || non_cache::within (non_cache_cnn.cpp:74-76)"""
        coords, _, _ = vina.set_conf(self.lig, conf)
        heavy = coords[self.smt > 1]
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
