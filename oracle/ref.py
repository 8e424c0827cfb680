"""ctypes wrapper over oracle/_ref/libgnina_ref.so = the REFERENCE's own Vina/smina CPU code compiled from
/root/reference/gninasrc/lib behind oracle/ref_shims/ (recipe: oracle/Makefile.ref, driver: oracle/ref_driver.cpp).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  It pins oracle/vina_ref.c (the restatement the GPU parity tests
use) and gnina_amd/host/pdbqt.cpp to the reference itself.

The library is built where /root/reference exists (this container); oracle/_ref/ is git-ignored but travels to the
GPU box with gpurun, where `available()` finds the prebuilt file."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libgnina_ref.so")
REFERENCE = "/root/reference"
_f32p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
_lib = None


def build(force=False):
    """make -f oracle/Makefile.ref; only possible where the reference sources are."""
    if not os.path.isdir(os.path.join(REFERENCE, "gninasrc", "lib")):
        return LIB if os.path.exists(LIB) else None
    cmd = ["make", "-f", os.path.join(HERE, "Makefile.ref"), "-j8", "REF=" + REFERENCE]
    if force:
        subprocess.run(cmd + ["clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    # the igrid drop-in test links the reference's objects with libmi_gnina.so: only once that exists
    if os.path.exists(os.path.join(os.path.dirname(HERE), "gnina_amd", "lib", "libmi_gnina.so")):
        subprocess.run(cmd + ["dropin", "dropin_cnn"], check=True, stdout=subprocess.DEVNULL)
    return LIB


def available():
    return os.path.exists(LIB) or build() is not None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.ref_last_error.restype = C.c_char_p
        L.ref_scene_create.restype = vp
        L.ref_scene_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, _f32p]
        L.ref_scene_free.argtypes = [vp]
        L.ref_sizes.argtypes = [vp, _i32p]
        L.ref_atoms.argtypes = [vp, _f32p, _i32p, _f32p]
        L.ref_grid_atoms.argtypes = [vp, _f32p, _i32p]
        L.ref_pairs.argtypes = [vp, C.c_int, _i32p, _i32p]
        L.ref_bonds.argtypes = [vp, C.c_int, _i32p, C.c_int]
        L.ref_table_n.argtypes = [vp]
        L.ref_table_eval.argtypes = [vp, C.c_int, C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p]
        L.ref_pair_energy.restype = C.c_float
        L.ref_pair_energy.argtypes = [vp, C.c_int, C.c_int, C.c_float]
        L.ref_build_grids.argtypes = [vp, _f32p, _f32p, C.c_float, C.c_int, _f32p, _f32p, _i32p, _i32p, C.c_int]
        L.ref_cache_probe.argtypes = [vp, C.c_int, _f32p, C.c_int, C.c_float, _f32p, _f32p]
        L.ref_set_user_grid.argtypes = [vp, _f32p, _f32p, _i32p, C.c_char_p, C.c_float]
        L.ref_set_approximation.argtypes = [vp, C.c_int, C.c_float]
        L.ref_set_line_search.argtypes = [vp, C.c_int]
        L.ref_prec_eval.argtypes = [vp, C.c_int, C.c_int, _f32p, C.c_int, _f32p, _f32p]
        L.ref_set_conf.argtypes = [vp, _f32p, _f32p]
        L.ref_initial_conf.argtypes = [vp, _f32p]
        L.ref_eval_deriv.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p]
        L.ref_eval.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.ref_ig_eval.argtypes = [vp, _f32p, C.c_float, C.c_int, _f32p]
        L.ref_final_energies.argtypes = [vp, _f32p, _f32p, _f32p, _f32p]
        L.ref_bfgs.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, _f32p, _f32p]
        L.ref_conf_increment.argtypes = [vp, _f32p, _f32p, C.c_float]
        L.ref_mc.argtypes = [vp, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _f32p, _f32p, C.c_int,
                             C.c_int, C.c_int, _f32p, _f32p, _f32p]
        L.ref_mc_parallel.argtypes = [vp, C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p,
                                      C.c_int, C.POINTER(C.c_double), _f32p]
        L.ref_mutate.argtypes = [vp, _f32p, C.c_uint, C.c_float]
        L.ref_within.argtypes = [vp, _f32p]
        L.ref_conf_independent.restype = C.c_float
        L.ref_conf_independent.argtypes = [vp, C.c_float]
        L.ref_random_stream.argtypes = [C.c_uint, C.c_int, _f32p, _i32p, _f32p]
        _lib = L
    return _lib


def _p(a, t=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class RefError(RuntimeError):
    pass


def _check(rc, bad=lambda r: r != 0):
    if bad(rc):
        raise RefError(lib().ref_last_error().decode())
    return rc


class Scene:
    """parse_receptor_pdbqt(rigid[, flex]) + m.append(parse_ligand_pdbqt(ligand)) + gnina's default scoring terms."""

    def __init__(self, rigid_text, ligand_text=None, flex_text=None, weights=None):
        w = None if weights is None else _f(weights)
        enc = lambda s: None if s is None else s.encode()
        self.h = lib().ref_scene_create(enc(rigid_text or ""), enc(flex_text), enc(ligand_text), _p(w))
        if not self.h:
            raise RefError(lib().ref_last_error().decode())
        s = np.zeros(10, dtype=np.int32)
        _check(lib().ref_sizes(self.h, _p(s, C.c_int32)))
        (self.n_atoms, self.n_movable, self.n_grid_atoms, self.n_lig_tors, self.n_flex, self.n_flex_tors,
         self.n_lig_pairs, self.n_other_pairs, self.lig_begin, self.lig_end) = (int(v) for v in s)
        self.has_ligand = ligand_text is not None
        self.conf_len = (7 + self.n_lig_tors if self.has_ligand else 0) + self.n_flex_tors
        self.change_len = (6 + self.n_lig_tors if self.has_ligand else 0) + self.n_flex_tors

    def __del__(self):
        try:
            lib().ref_scene_free(self.h)
        except Exception:
            pass

    def atoms(self):
        xyz = np.zeros((self.n_atoms, 3), dtype=np.float32)
        internal = np.zeros((self.n_atoms, 3), dtype=np.float32)
        smt = np.zeros(self.n_atoms, dtype=np.int32)
        _check(lib().ref_atoms(self.h, _p(xyz), _p(smt, C.c_int32), _p(internal)))
        return xyz, smt, internal

    def grid_atoms(self):
        xyz = np.zeros((self.n_grid_atoms, 3), dtype=np.float32)
        smt = np.zeros(self.n_grid_atoms, dtype=np.int32)
        _check(lib().ref_grid_atoms(self.h, _p(xyz), _p(smt, C.c_int32)))
        return xyz, smt

    def pairs(self, other=False):
        n = self.n_other_pairs if other else self.n_lig_pairs
        out = np.zeros((n, 2), dtype=np.int32)
        t12 = np.zeros((n, 2), dtype=np.int32)
        _check(lib().ref_pairs(self.h, int(other), _p(out, C.c_int32), _p(t12, C.c_int32)))
        return out, t12

    def bonds(self, i):
        """bond list of atom i of the combined index space (grid_atoms first), in the reference's list order"""
        out = np.zeros(16, dtype=np.int32)
        n = _check(lib().ref_bonds(self.h, int(i), _p(out, C.c_int32), 16), lambda r: r < 0)
        return out[:n].tolist()

    def table_n(self):
        return lib().ref_table_n(self.h)

    def table_eval(self, t1, t2, r2):
        r2 = _f(r2)
        fast, e, dor = (np.zeros(len(r2), dtype=np.float32) for _ in range(3))
        _check(lib().ref_table_eval(self.h, t1, t2, _p(r2), len(r2), _p(fast), _p(e), _p(dor)))
        return fast, e, dor

    def pair_energy(self, t1, t2, r):
        return lib().ref_pair_energy(self.h, t1, t2, r)

    def build_grids(self, center, size, slope=1e3, build_cache=True, extra_types=()):
        b, e, n = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.int32)
        _check(lib().ref_build_grids(self.h, _p(_f(center)), _p(_f(size)), slope, int(build_cache), _p(b), _p(e),
                                     _p(n, C.c_int32), _p(np.ascontiguousarray(extra_types, dtype=np.int32), C.c_int32),
                                     len(extra_types)))
        self.gd = (b, e, n)
        return b, e, n

    def set_user_grid(self, begin, end, n, value_lines, scale=1.0):
        """grid::init(gd, user_in, scale) on the file's value lines (one number per line); None removes the grid.
        Call before build_grids: cache::populate bakes it into the lattice."""
        if value_lines is None:
            _check(lib().ref_set_user_grid(self.h, None, None, None, None, 1.0))
            return
        nn = np.ascontiguousarray(n, dtype=np.int32)
        _check(lib().ref_set_user_grid(self.h, _p(_f(begin)), _p(_f(end)), _p(nn, C.c_int32),
                                       value_lines.encode() if isinstance(value_lines, str) else value_lines,
                                       float(scale)))

    def set_approximation(self, kind, factor=10.0):
        """--approximation: 0 = precalculate_linear(wt, 32), 1 = precalculate_splines(wt, factor); before build_grids"""
        _check(lib().ref_set_approximation(self.h, int(kind), float(factor)))

    def set_line_search(self, accurate, simple=False):
        """--accurate_line_search (or, simple: --simple_ascent = minimization_params::Simple) for bfgs() and mc()"""
        _check(lib().ref_set_line_search(self.h, 2 if simple else 1 if accurate else 0))

    def prec_eval(self, t1, t2, r2):
        """(E, dE/dr / r) of the run's precalculate for a type pair at squared distances r2"""
        r2 = _f(r2)
        e, d = np.zeros(len(r2), np.float32), np.zeros(len(r2), np.float32)
        _check(lib().ref_prec_eval(self.h, int(t1), int(t2), _p(r2), len(r2), _p(e), _p(d)))
        return e, d

    def cache_probe(self, t, xyz, v=1000.0, deriv=False):
        xyz = _f(xyz).reshape(-1, 3)
        e = np.zeros(len(xyz), dtype=np.float32)
        d = np.zeros((len(xyz), 3), dtype=np.float32) if deriv else None
        _check(lib().ref_cache_probe(self.h, int(t), _p(xyz), len(xyz), v, _p(e), _p(d)))
        return (e, d) if deriv else e

    def cache_grid(self, t):
        """The populated grid of ligand type t, read back at its lattice points: [nz+1][ny+1][nx+1], x fastest."""
        b, e, n = self.gd
        ax = [b[i] + (e[i] - b[i]) * np.arange(n[i] + 1, dtype=np.float32) / np.float32(n[i]) for i in range(3)]
        return ax

    def initial_conf(self):
        x = np.zeros(self.conf_len, dtype=np.float32)
        _check(lib().ref_initial_conf(self.h, _p(x)))
        return x

    def set_conf(self, conf):
        xyz = np.zeros((self.n_atoms, 3), dtype=np.float32)
        _check(lib().ref_set_conf(self.h, _p(_f(conf)), _p(xyz)))
        return xyz

    def eval_deriv(self, conf, v=(1000.0, 1000.0, 1000.0), ig=0, prec=0):
        """model::eval_deriv -> (energy, change, coords, minus_forces); ig 0 cache / 1 non_cache; prec 0 linear / 1 exact"""
        e = C.c_float()
        chg = np.zeros(self.change_len, dtype=np.float32)
        xyz = np.zeros((self.n_atoms, 3), dtype=np.float32)
        mf = np.zeros((self.n_atoms, 3), dtype=np.float32)
        _check(lib().ref_eval_deriv(self.h, _p(_f(conf)), _p(_f(v)), ig, prec, C.byref(e), _p(chg), _p(xyz), _p(mf)))
        return e.value, chg, xyz, mf

    def eval(self, conf, v=(1000.0, 1000.0, 1000.0), ig=0, prec=0):
        e = C.c_float()
        _check(lib().ref_eval(self.h, _p(_f(conf)), _p(_f(v)), ig, prec, C.byref(e)))
        return e.value

    def ig_eval(self, conf, v=1000.0, ig=0):
        e = C.c_float()
        _check(lib().ref_ig_eval(self.h, _p(_f(conf)), v, ig, C.byref(e)))
        return e.value

    def final_energies(self, conf, v=(1000.0, 1000.0, 1000.0)):
        e, intra = C.c_float(), C.c_float()
        _check(lib().ref_final_energies(self.h, _p(_f(conf)), _p(_f(v)), C.byref(e), C.byref(intra)))
        return e.value, intra.value

    def bfgs(self, conf, v=(1000.0, 1000.0, 1000.0), ig=0, max_iters=None):
        if max_iters is None:
            max_iters = (25 + self.n_movable) // 3
        x = np.array(conf, dtype=np.float32, copy=True)
        e = C.c_float()
        chg = np.zeros(self.change_len, dtype=np.float32)
        _check(lib().ref_bfgs(self.h, _p(x), _p(_f(v)), ig, int(max_iters), C.byref(e), _p(chg)))
        return e.value, x, chg

    def conf_increment(self, conf, change, alpha):
        x = np.array(conf, dtype=np.float32, copy=True)
        _check(lib().ref_conf_increment(self.h, _p(x), _p(_f(change)), alpha))
        return x

    def mc(self, seed, n_steps, corner1, corner2, max_iters=None, num_saved=50, temperature=1.2, min_rmsd=1.0, ig=0,
           n_heavy=None):
        if max_iters is None:
            max_iters = (25 + self.n_movable) // 3
        _, smt, _ = self.atoms()
        if n_heavy is None:
            n_heavy = int((smt[:self.n_movable] > 1).sum())
        e = np.zeros(num_saved, dtype=np.float32)
        cf = np.zeros((num_saved, self.conf_len), dtype=np.float32)
        xyz = np.zeros((num_saved, n_heavy, 3), dtype=np.float32)
        n = _check(lib().ref_mc(self.h, int(seed), int(n_steps), int(max_iters), int(num_saved), temperature, min_rmsd,
                                _p(_f(corner1)), _p(_f(corner2)), ig, self.conf_len, n_heavy, _p(e), _p(cf), _p(xyz)),
                   lambda r: r < 0)
        return e[:n], cf[:n], xyz[:n]

    def mc_parallel(self, seeds, n_threads, n_steps, corner1, corner2, max_iters=None, num_saved=50, ig=0):
        """parallel_mc's fan-out (parallel_mc.cpp:183-214): len(seeds) chains, private model copies, n_threads workers
        over the shared cache.  Returns (wall seconds, best energy per chain)."""
        if max_iters is None:
            max_iters = (25 + self.n_movable) // 3
        sd = np.ascontiguousarray(seeds, dtype=np.uint32)
        best = np.zeros(len(sd), dtype=np.float32)
        sec = C.c_double(0.0)
        _check(lib().ref_mc_parallel(self.h, sd.ctypes.data_as(C.POINTER(C.c_uint)), len(sd), int(n_threads), int(n_steps),
                                     int(max_iters), int(num_saved), _p(_f(corner1)), _p(_f(corner2)), ig, C.byref(sec),
                                     _p(best)))
        return sec.value, best

    def mutate(self, conf, seed, amplitude=2.0):
        x = np.array(conf, dtype=np.float32, copy=True)
        _check(lib().ref_mutate(self.h, _p(x), int(seed), amplitude))
        return x

    def within(self, conf):
        return bool(_check(lib().ref_within(self.h, _p(_f(conf))), lambda r: r < 0))

    def conf_independent(self, e):
        return lib().ref_conf_independent(self.h, e)


def random_stream(seed, n):
    u, i, g = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float32)
    lib().ref_random_stream(int(seed), n, _p(u), _p(i, C.c_int32), _p(g))
    return u, i, g
