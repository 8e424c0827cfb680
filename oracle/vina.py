"""ctypes wrapper over oracle/vina_ref.c (CPU restatement of the smina/Vina scoring + BFGS path).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import ctypes as C

import numpy as np

from . import voxel as _voxel

_f32p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)


class GridDims(C.Structure):
    _fields_ = [("begin", C.c_float * 3), ("end", C.c_float * 3), ("n", C.c_int * 3)]

    @property
    def shape(self):  # (nz+1, ny+1, nx+1): x fastest in memory
        return (self.n[2] + 1, self.n[1] + 1, self.n[0] + 1)


class Ligand(C.Structure):
    _fields_ = [("n_atoms", C.c_int), ("smt", _i32p), ("local_xyz", _f32p), ("n_nodes", C.c_int),
                ("parent", _i32p), ("abeg", _i32p), ("aend", _i32p), ("rel_origin", _f32p), ("rel_axis", _f32p),
                ("n_pairs", C.c_int), ("pairs", _i32p),
                # flexible residues (see ora_ligand in vina_ref.c): 0 / NULL = a plain ligand
                ("n_movable", C.c_int), ("pair_kind", _i32p), ("lig_begin", C.c_int), ("lig_end", C.c_int)]


_configured = False


def lib():
    global _configured
    L = _voxel.lib()
    if not _configured:
        vp = C.c_void_p
        L.ora_vina_pair_energy.restype = C.c_float
        L.ora_vina_pair_energy.argtypes = [_f32p, C.c_int, C.c_int, C.c_float]
        L.ora_vina_tables_create.restype = vp
        L.ora_vina_tables_create.argtypes = [_f32p, C.c_float, C.c_float]
        L.ora_vina_tables_free.argtypes = [vp]
        L.ora_vina_tables_n.restype = C.c_int
        L.ora_vina_tables_n.argtypes = [vp]
        L.ora_vina_tables_get.argtypes = [vp, C.c_int, C.c_int, _f32p, _f32p, _f32p]
        L.ora_vina_eval_fast.restype = C.c_float
        L.ora_vina_eval_fast.argtypes = [vp, C.c_int, C.c_int, C.c_float]
        L.ora_vina_table_eval_deriv.restype = None
        L.ora_vina_table_eval_deriv.argtypes = [vp, C.c_int, C.c_int, C.c_float, _f32p, _f32p]
        L.ora_vina_setup_grid_dims.argtypes = [_f32p, _f32p, C.POINTER(GridDims)]
        L.ora_vina_cache_populate.argtypes = [vp, C.POINTER(GridDims), _f32p, _i32p, C.c_int, C.c_int, _f32p]
        L.ora_vina_grid_evaluate.restype = C.c_float
        L.ora_vina_grid_evaluate.argtypes = [C.POINTER(GridDims), _f32p, _f32p, C.c_float, C.c_float, _f32p]
        L.ora_vina_set_conf.argtypes = [C.POINTER(Ligand), _f32p, _f32p, _f32p, _f32p]
        L.ora_vina_conf_increment.argtypes = [_f32p, _f32p, C.c_float, C.c_int]
        _configured = True
    return L


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class Tables:
    def __init__(self, weights=None, cutoff=8.0, factor=32.0):
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        self.h = lib().ora_vina_tables_create(_p(w) if w is not None else None, cutoff, factor)
        self.n = lib().ora_vina_tables_n(self.h)

    def get(self, t1, t2):
        fast, se, sd = (np.empty(self.n, dtype=np.float32) for _ in range(3))
        lib().ora_vina_tables_get(self.h, t1, t2, _p(fast), _p(se), _p(sd))
        return fast, se, sd

    def eval_fast(self, t1, t2, r2):
        return lib().ora_vina_eval_fast(self.h, t1, t2, r2)

    def eval_deriv(self, t1, t2, r2):
        e, d = C.c_float(), C.c_float()
        lib().ora_vina_table_eval_deriv(self.h, t1, t2, r2, C.byref(e), C.byref(d))
        return e.value, d.value

    def __del__(self):
        try:
            lib().ora_vina_tables_free(self.h)
        except Exception:
            pass


def pair_energy(t1, t2, r, weights=None):
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    return lib().ora_vina_pair_energy(_p(w) if w is not None else None, t1, t2, r)


def setup_grid_dims(center, size):
    gd = GridDims()
    c = np.ascontiguousarray(center, dtype=np.float32)
    s = np.ascontiguousarray(size, dtype=np.float32)
    lib().ora_vina_setup_grid_dims(_p(c), _p(s), C.byref(gd))
    return gd


def cache_populate(tables, gd, rec_xyz, rec_smt, lig_type):
    rec_xyz = np.ascontiguousarray(rec_xyz, dtype=np.float32)
    rec_smt = np.ascontiguousarray(rec_smt, dtype=np.int32)
    out = np.empty(gd.shape, dtype=np.float32)
    lib().ora_vina_cache_populate(tables.h, C.byref(gd), _p(rec_xyz), _p(rec_smt, C.c_int32), len(rec_smt),
                                  int(lig_type), _p(out))
    return out


def grid_evaluate(gd, data, loc, slope, v, deriv=True):
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    d = np.zeros(3, dtype=np.float32)
    e = lib().ora_vina_grid_evaluate(C.byref(gd), _p(data), _p(loc), slope, v, _p(d) if deriv else None)
    return e, d


class LigandHandle:
    """Keeps the numpy arrays alive behind the C struct."""

    def __init__(self, lig):
        self.arr = {k: np.ascontiguousarray(lig[k]) for k in
                    ("smt", "local_xyz", "parent", "abeg", "aend", "rel_origin", "rel_axis", "pairs")}
        a = self.arr
        self.n_atoms, self.n_nodes = len(a["smt"]), len(a["parent"])
        self.n_tors = self.n_nodes - 1
        if lig.get("pair_kind") is not None:
            a["pair_kind"] = np.ascontiguousarray(lig["pair_kind"], dtype=np.int32)
        self.n_movable = int(lig.get("n_movable", 0)) or self.n_atoms
        self.c = Ligand(self.n_atoms, _p(a["smt"], C.c_int32), _p(a["local_xyz"]), self.n_nodes,
                        _p(a["parent"], C.c_int32), _p(a["abeg"], C.c_int32), _p(a["aend"], C.c_int32),
                        _p(a["rel_origin"]), _p(a["rel_axis"]), len(a["pairs"]), _p(a["pairs"], C.c_int32),
                        int(lig.get("n_movable", 0)), _p(a["pair_kind"], C.c_int32) if "pair_kind" in a else None,
                        int(lig.get("lig_begin", 0)), int(lig.get("lig_end", 0)))


def set_conf(lig, conf):
    conf = np.ascontiguousarray(conf, dtype=np.float32)
    coords = np.empty((lig.n_atoms, 3), dtype=np.float32)
    origin = np.empty((lig.n_nodes, 3), dtype=np.float32)
    axis = np.empty((lig.n_nodes, 3), dtype=np.float32)
    lib().ora_vina_set_conf(C.byref(lig.c), _p(conf), _p(coords), _p(origin), _p(axis))
    return coords, origin, axis


class Scene:
    """tables + grid dims + per-type grids + ligand: everything model::eval_deriv needs."""

    def __init__(self, tables, gd, grids_by_type, lig, slope=1e3):
        self.tables, self.gd, self.lig, self.slope = tables, gd, lig, slope
        self.grids = grids_by_type
        self.ptrs = (_f32p * 28)()
        for t in range(28):
            self.ptrs[t] = _p(grids_by_type[t]) if t in grids_by_type else None
        lib()

    def eval_deriv(self, conf, v=(1000.0, 1000.0, 1000.0)):
        """model::eval_deriv -> (energy, change[6+T], coords, forces)"""
        conf = np.ascontiguousarray(conf, dtype=np.float32)
        vv = np.ascontiguousarray(v, dtype=np.float32)
        change = np.empty(6 + self.lig.n_tors, dtype=np.float32)
        coords = np.empty((self.lig.n_atoms, 3), dtype=np.float32)
        forces = np.empty((self.lig.n_atoms, 3), dtype=np.float32)
        f = _model_eval_deriv()
        e = f(self.tables.h, C.byref(self.gd), self.ptrs, self.slope, C.byref(self.lig.c), _p(conf), _p(vv),
              _p(change), _p(coords), _p(forces))
        return e, change, coords, forces

    def eval(self, conf, v=(1000.0, 1000.0, 1000.0)):
        conf = np.ascontiguousarray(conf, dtype=np.float32)
        vv = np.ascontiguousarray(v, dtype=np.float32)
        f = _voxel.lib().ora_vina_eval
        f.restype = C.c_float
        f.argtypes = [C.c_void_p, C.POINTER(GridDims), C.POINTER(_f32p), C.c_float, C.POINTER(Ligand), _f32p, _f32p]
        return f(self.tables.h, C.byref(self.gd), self.ptrs, self.slope, C.byref(self.lig.c), _p(conf), _p(vv))

    def bfgs(self, conf, v=(1000.0, 1000.0, 1000.0), max_iters=None):
        """quasi_newton: returns (energy, conf_out, grad, n_evals)"""
        if max_iters is None:
            max_iters = (25 + self.lig.n_atoms) // 3  # main.cpp:454-456
        conf = np.array(conf, dtype=np.float32, copy=True)
        vv = np.ascontiguousarray(v, dtype=np.float32)
        g = np.empty(6 + self.lig.n_tors, dtype=np.float32)
        ev = C.c_long()
        f = _voxel.lib().ora_vina_bfgs
        f.restype = C.c_float
        f.argtypes = [C.c_void_p, C.POINTER(GridDims), C.POINTER(_f32p), C.c_float, C.POINTER(Ligand), _f32p, _f32p,
                      C.c_int, _f32p, C.POINTER(C.c_long)]
        e = f(self.tables.h, C.byref(self.gd), self.ptrs, self.slope, C.byref(self.lig.c), _p(conf), _p(vv),
              int(max_iters), _p(g), C.byref(ev))
        return e, conf, g, ev.value


class McParams(C.Structure):
    _fields_ = [("n_steps", C.c_int), ("max_iters", C.c_int), ("num_saved", C.c_int), ("temperature", C.c_float),
                ("mutation_amplitude", C.c_float), ("min_rmsd", C.c_float), ("hunt_cap", C.c_float * 3),
                ("authentic_v", C.c_float * 3)]


def model_energies(scene, rec_xyz, rec_smt, conf, v=(1000.0, 1000.0, 1000.0), slope=None):
    """do_search's two energies for any model, flexible residues included (main.cpp:339-344): (model::eval(exact_prec,
    non_cache) before conf_independent, eval_intramolecular(exact_prec))"""
    conf = np.ascontiguousarray(conf, dtype=np.float32)
    vv = np.ascontiguousarray(v, dtype=np.float32)
    rec_xyz = np.ascontiguousarray(rec_xyz, dtype=np.float32)
    rec_smt = np.ascontiguousarray(rec_smt, dtype=np.int32)
    intra = C.c_float()
    f = _voxel.lib().ora_vina_model_energies
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, _f32p, C.POINTER(GridDims), C.c_float, _f32p, _i32p, C.c_int, C.POINTER(Ligand), _f32p, _f32p,
                  C.POINTER(C.c_float)]
    e = f(scene.tables.h, None, C.byref(scene.gd), scene.slope if slope is None else slope, _p(rec_xyz),
          _p(rec_smt, C.c_int32), len(rec_smt), C.byref(scene.lig.c), _p(conf), _p(vv), C.byref(intra))
    return e, intra.value


def set_line_search(accurate=False, simple=False):
    """--accurate_line_search: every bfgs of this library (bfgs, bfgs_callback, refine, mc) then runs
    accurate_line_search (bfgs.h:104-180); simple: --simple_ascent, simple_gradient_ascent (bfgs.h:234-355) instead of
    bfgs<>.  Process-wide: tests reset it."""
    f = _voxel.lib().ora_vina_set_line_search
    f.restype = None
    f.argtypes = [C.c_int]
    f(2 if simple else 1 if accurate else 0)


def set_approximation(kind=0, factor=10.0, cutoff=8.0, weights=None):
    """--approximation: 0 = precalculate_linear (the tables), 1 = precalculate_splines(sf, factor): every table look-up of
    this library (cache_populate, eval, eval_deriv, non_cache, bfgs, mc) then evaluates the pair's spline.
    Process-wide: tests reset it."""
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    f = _voxel.lib().ora_vina_set_approximation
    f.restype = None
    f.argtypes = [C.c_int, C.c_float, C.c_float, _f32p]
    f(int(kind), float(factor), float(cutoff), None if w is None else _p(w))


def user_grid_data(begin, end, n, values, scale=1.0):
    """grid::init(gd, user_in, scale) (grid.cpp:69-92) -> (GridDims, data[(nz+1)][(ny+1)][(nx+1)]): the file fills
    [0, n)^3 with -(value * scale), the last plane of every dimension stays 0."""
    gd = GridDims()
    for i in range(3):
        gd.begin[i], gd.end[i], gd.n[i] = float(begin[i]), float(end[i]), int(n[i])
    data = np.zeros(gd.shape, dtype=np.float32)
    v = np.asarray(values, dtype=np.float64).reshape(int(n[2]), int(n[1]), int(n[0]))
    data[:n[2], :n[1], :n[0]] = (-(v * float(np.float32(scale)))).astype(np.float32)
    return gd, data


def set_user_grid(begin=None, end=None, n=None, values=None, scale=1.0, cache_slope=1e3):
    """--user_grid for the functions of this library that see it in the reference: cache_populate (at lattice
    indices, cache.cpp:177-179) and noncache_eval with derivatives (non_cache.cpp:168-173).  values [nz][ny][nx]
    float64; None removes it.  Process-wide: tests reset it."""
    f = _voxel.lib().ora_vina_set_user_grid
    f.restype = None
    f.argtypes = [_f32p, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_float, C.c_float]
    if values is None:
        f(None, None, None, None, 1.0, 1e3)
        return
    b = np.ascontiguousarray(begin, dtype=np.float32)
    e = np.ascontiguousarray(end, dtype=np.float32)
    nn = np.ascontiguousarray(n, dtype=np.int32)
    v = np.ascontiguousarray(values, dtype=np.float64).ravel()
    f(_p(b), _p(e), nn.ctypes.data_as(C.POINTER(C.c_int)), v.ctypes.data_as(C.POINTER(C.c_double)), float(scale),
      float(cache_slope))


def mc_chain(scene, corner1, corner2, seed, n_steps, max_iters, num_saved=50, temperature=1.2, amplitude=2.0,
             min_rmsd=1.0, rng_kind=1, conf0=None):
    """monte_carlo::operator() for one chain -> (energies [n], confs [n,7+T], coords [n,nh,3], evals).
    rng_kind 1 (default, also what the HIP kernel draws from): mt19937 + Boost's distributions as restated for
    oracle/_ref -- the chain follows the reference's step for step; 0: a counter-based splitmix stream (legacy)."""
    lig = scene.lig
    nh = int((lig.arr["smt"][:lig.n_movable] > 1).sum())
    P = McParams(n_steps, max_iters, num_saved, temperature, amplitude, min_rmsd, (C.c_float * 3)(10, 10, 10),
                 (C.c_float * 3)(1000, 1000, 1000))
    e = np.zeros(num_saved, dtype=np.float32)
    cf = np.zeros((num_saved, 7 + lig.n_tors), dtype=np.float32)
    xyz = np.zeros((num_saved, nh, 3), dtype=np.float32)
    c1 = np.ascontiguousarray(corner1, dtype=np.float32)
    c2 = np.ascontiguousarray(corner2, dtype=np.float32)
    ev = C.c_long()
    f = _voxel.lib().ora_vina_mc_chain_rng
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(GridDims), C.POINTER(_f32p), C.c_float, C.POINTER(Ligand), _f32p, _f32p,
                  C.c_uint64, C.c_int, _f32p, C.POINTER(McParams), _f32p, _f32p, _f32p, C.POINTER(C.c_long)]
    c0 = None if conf0 is None else np.ascontiguousarray(conf0, dtype=np.float32)
    n = f(scene.tables.h, C.byref(scene.gd), scene.ptrs, scene.slope, C.byref(lig.c), _p(c1), _p(c2), int(seed),
          int(rng_kind), None if c0 is None else _p(c0), C.byref(P), _p(e), _p(cf), _p(xyz), C.byref(ev))
    return e[:n], cf[:n], xyz[:n], ev.value


def noncache_eval(scene, rec_xyz, rec_smt, conf, v=(1000.0, 1000.0, 1000.0), deriv=True, exact=False, slope=None):
    """model::eval_deriv / model::eval with ig = non_cache -> (total, change, inter, intra)"""
    conf = np.ascontiguousarray(conf, dtype=np.float32)
    vv = np.ascontiguousarray(v, dtype=np.float32)
    rec_xyz = np.ascontiguousarray(rec_xyz, dtype=np.float32)
    rec_smt = np.ascontiguousarray(rec_smt, dtype=np.int32)
    change = np.zeros(6 + scene.lig.n_tors, dtype=np.float32)
    inter, intra = C.c_float(), C.c_float()
    f = _voxel.lib().ora_vina_noncache_eval
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, _f32p, C.c_int, C.POINTER(GridDims), C.c_float, _f32p, _i32p, C.c_int,
                  C.POINTER(Ligand), _f32p, _f32p, C.c_int, _f32p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    e = f(scene.tables.h, None, int(exact), C.byref(scene.gd), scene.slope if slope is None else slope, _p(rec_xyz),
          _p(rec_smt, C.c_int32), len(rec_smt), C.byref(scene.lig.c), _p(conf), _p(vv), int(deriv), _p(change),
          C.byref(inter), C.byref(intra))
    return e, change, inter.value, intra.value


def refine(scene, rec_xyz, rec_smt, conf, v=(1000.0, 1000.0, 1000.0), max_iters=None):
    """refine_structure -> (energy, conf, tries)"""
    conf = np.array(conf, dtype=np.float32, copy=True)
    vv = np.ascontiguousarray(v, dtype=np.float32)
    rec_xyz = np.ascontiguousarray(rec_xyz, dtype=np.float32)
    rec_smt = np.ascontiguousarray(rec_smt, dtype=np.int32)
    if max_iters is None:
        max_iters = (25 + scene.lig.n_atoms) // 3
    tries = C.c_int()
    f = _voxel.lib().ora_vina_refine
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, C.POINTER(GridDims), _f32p, _i32p, C.c_int, C.POINTER(Ligand), _f32p, _f32p, C.c_int,
                  C.POINTER(C.c_int)]
    e = f(scene.tables.h, C.byref(scene.gd), _p(rec_xyz), _p(rec_smt, C.c_int32), len(rec_smt), C.byref(scene.lig.c),
          _p(conf), _p(vv), int(max_iters), C.byref(tries))
    return e, conf, tries.value


def conf_independent(e, num_tors):
    f = _voxel.lib().ora_vina_conf_independent
    f.restype = C.c_float
    f.argtypes = [C.c_float, C.c_float]
    return f(e, num_tors)


def cache_eval(scene, conf, v1=1000.0):
    conf = np.ascontiguousarray(conf, dtype=np.float32)
    f = _voxel.lib().ora_vina_cache_eval
    f.restype = C.c_float
    f.argtypes = [C.POINTER(GridDims), C.POINTER(_f32p), C.c_float, C.POINTER(Ligand), _f32p, C.c_float]
    return f(C.byref(scene.gd), scene.ptrs, scene.slope, C.byref(scene.lig.c), _p(conf), v1)


def _model_eval_deriv():
    f = _voxel.lib().ora_vina_model_eval_deriv
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, C.POINTER(GridDims), C.POINTER(_f32p), C.c_float, C.POINTER(Ligand), _f32p, _f32p,
                  _f32p, _f32p, _f32p]
    return f


def conf_increment(conf, p, alpha, n_tors):
    conf = np.array(conf, dtype=np.float32, copy=True)
    p = np.ascontiguousarray(p, dtype=np.float32)
    lib().ora_vina_conf_increment(_p(conf), _p(p), alpha, n_tors)
    return conf


def forces_to_change(lig, conf, forces):
    """ligands.derivative on given per-atom forces -> (change [6+T], coords [n_atoms, 3])"""
    conf = np.ascontiguousarray(conf, dtype=np.float32)
    forces = np.ascontiguousarray(forces, dtype=np.float32)
    change = np.zeros(6 + lig.n_tors, dtype=np.float32)
    coords = np.empty((lig.n_atoms, 3), dtype=np.float32)
    f = _voxel.lib().ora_vina_forces_to_change
    f.restype = None
    f.argtypes = [C.POINTER(Ligand), _f32p, _f32p, _f32p, _f32p]
    f(C.byref(lig.c), _p(conf), _p(forces), _p(change), _p(coords))
    return change, coords


_FX = C.CFUNCTYPE(C.c_float, _f32p, _f32p, C.c_void_p)


def bfgs_callback(lig, conf, fx, max_iters):
    """quasi_newton on a Python objective fx(conf) -> (energy, change): (energy, conf, grad, evals)"""
    conf = np.array(conf, dtype=np.float32, copy=True)
    n, nc = 6 + lig.n_tors, 7 + lig.n_tors

    def cb(cp, gp, _user):
        c = np.ctypeslib.as_array(cp, shape=(nc,)).copy()
        e, g = fx(c)
        np.ctypeslib.as_array(gp, shape=(n,))[:] = np.asarray(g, dtype=np.float32)
        return float(e)

    g = np.empty(n, dtype=np.float32)
    ev = C.c_long()
    f = _voxel.lib().ora_vina_bfgs_cb
    f.restype = C.c_float
    f.argtypes = [C.POINTER(Ligand), _f32p, C.c_int, _FX, C.c_void_p, _f32p, C.POINTER(C.c_long)]
    keep = _FX(cb)
    e = f(C.byref(lig.c), _p(conf), int(max_iters), keep, None, _p(g), C.byref(ev))
    return e, conf, g, ev.value


def mutate(lig, conf, seed, amplitude=2.0, rng_kind=1):
    """mutate_conf (mutate.cpp:35-73) with a freshly seeded generator"""
    x = np.array(conf, dtype=np.float32, copy=True)
    f = _voxel.lib().ora_vina_mutate
    f.restype = None
    f.argtypes = [C.POINTER(Ligand), C.c_int, C.c_uint64, C.c_float, _f32p]
    f(C.byref(lig.c), int(rng_kind), int(seed), amplitude, _p(x))
    return x


def random_stream(kind, seed, n):
    u, i, g = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float32)
    f = _voxel.lib().ora_vina_random_stream
    f.restype = None
    f.argtypes = [C.c_int, C.c_uint64, C.c_int, _f32p, _i32p, _f32p]
    f(int(kind), int(seed), n, _p(u), _p(i, C.c_int32), _p(g))
    return u, i, g
