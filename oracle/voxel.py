"""ctypes wrapper over oracle/voxel_ref.c (the CPU restatement of libmolgrid's voxelizer).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Each function mirrors one C entry point;
the C file cites the reference call sites (gninasrc/lib/torch_model.cpp:108-181,200-206).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NUM_SMINA_TYPES = 28


def build():
    """Compile liboracle.so with oracle/Makefile (gcc, no FMA contraction)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.ora_smina_type_name.restype = C.c_char_p
        L.ora_smina_type_name.argtypes = [C.c_int]
        L.ora_smina_xs_radius.restype = C.c_float
        L.ora_smina_xs_radius.argtypes = [C.c_int]
        L.ora_typer_parse.restype = C.c_int
        L.ora_typer_parse.argtypes = [C.c_char_p, i32p]
        L.ora_type_atoms.restype = None
        L.ora_type_atoms.argtypes = [i32p, C.c_int, i32p, i32p, f32p]
        L.ora_center.restype = None
        L.ora_center.argtypes = [f32p, i32p, C.c_int, C.c_int, f32p]
        L.ora_grid_points.restype = C.c_int
        L.ora_grid_points.argtypes = [C.c_float, C.c_float]
        L.ora_grid_forward.restype = None
        L.ora_grid_forward.argtypes = [f32p, f32p, i32p, f32p, C.c_int, C.c_int, C.c_float, C.c_float,
                                       C.c_float, C.c_int, f32p]
        L.ora_grid_backward.restype = None
        L.ora_grid_backward.argtypes = [f32p, f32p, i32p, f32p, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_float, f32p, f32p]
        L.ora_voxelize_pose.restype = C.c_int
        L.ora_voxelize_pose.argtypes = [f32p, i32p, C.c_int, f32p, i32p, C.c_int, i32p, C.c_int, i32p, C.c_int,
                                        f32p, C.c_float, C.c_float, C.c_float, f32p, f32p]
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def smina_type_names():
    return [lib().ora_smina_type_name(i).decode() for i in range(NUM_SMINA_TYPES)]


def xs_radii():
    return np.array([lib().ora_smina_xs_radius(i) for i in range(NUM_SMINA_TYPES)], dtype=np.float32)


def typer_parse(map_text):
    """FileMappedGninaTyper: returns (chan_of_smt[28] int32, n_channels)."""
    out = np.full(NUM_SMINA_TYPES, -1, dtype=np.int32)
    n = lib().ora_typer_parse(map_text.encode(), _p(out, C.c_int32))
    if n < 0:
        raise ValueError("unknown smina type name in map")
    return out, n


def type_atoms(smt, chan_of_smt):
    smt = _i32(smt)
    chan = np.empty(len(smt), dtype=np.int32)
    rad = np.empty(len(smt), dtype=np.float32)
    cm = _i32(chan_of_smt)
    lib().ora_type_atoms(_p(smt, C.c_int32), len(smt), _p(cm, C.c_int32), _p(chan, C.c_int32), _p(rad, C.c_float))
    return chan, rad


def center(xyz, chan=None, only_typed=False):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty(3, dtype=np.float32)
    ch = _i32(chan) if chan is not None else None
    lib().ora_center(_p(xyz, C.c_float), _p(ch, C.c_int32) if ch is not None else None, len(xyz),
                     int(only_typed), _p(out, C.c_float))
    return out


def grid_points(resolution, dimension):
    return lib().ora_grid_points(resolution, dimension)


def grid_forward(center_xyz, xyz, chan, radius, n_channels, resolution=0.5, dimension=23.5, radius_scale=1.0,
                 binary=False, out=None):
    """GridMaker::forward: returns float32 [n_channels, N, N, N] (x slowest, z fastest)."""
    xyz = _f32(xyz).reshape(-1, 3)
    chan = _i32(chan)
    radius = _f32(radius)
    c = _f32(center_xyz)
    N = grid_points(resolution, dimension)
    if out is None:
        out = np.zeros((n_channels, N, N, N), dtype=np.float32)
    lib().ora_grid_forward(_p(c, C.c_float), _p(xyz, C.c_float), _p(chan, C.c_int32), _p(radius, C.c_float),
                           len(xyz), n_channels, resolution, dimension, radius_scale, int(binary),
                           _p(out, C.c_float))
    return out


def grid_backward(center_xyz, xyz, chan, radius, n_channels, gridgrad, resolution=0.5, dimension=23.5,
                  radius_scale=1.0):
    xyz = _f32(xyz).reshape(-1, 3)
    chan = _i32(chan)
    radius = _f32(radius)
    c = _f32(center_xyz)
    gg = _f32(gridgrad)
    out = np.zeros((len(xyz), 3), dtype=np.float32)
    lib().ora_grid_backward(_p(c, C.c_float), _p(xyz, C.c_float), _p(chan, C.c_int32), _p(radius, C.c_float),
                            len(xyz), n_channels, resolution, dimension, radius_scale, _p(gg, C.c_float),
                            _p(out, C.c_float))
    return out


def voxelize_pose(rec_xyz, rec_smt, lig_xyz, lig_smt, rec_map, lig_map, center_xyz=None, resolution=0.5,
                  dimension=23.5, radius_scale=1.0):
    """TorchModel::forward's voxelization half for one pose.  rec_map/lig_map = (chan_of_smt, n_ch).
    Returns (grid [C,N,N,N], center[3])."""
    rec_xyz = _f32(rec_xyz).reshape(-1, 3)
    lig_xyz = _f32(lig_xyz).reshape(-1, 3)
    rec_smt, lig_smt = _i32(rec_smt), _i32(lig_smt)
    rmap, nrc = _i32(rec_map[0]), int(rec_map[1])
    lmap, nlc = _i32(lig_map[0]), int(lig_map[1])
    N = grid_points(resolution, dimension)
    out = np.empty((nrc + nlc, N, N, N), dtype=np.float32)
    cen = np.empty(3, dtype=np.float32)
    cin = _f32(center_xyz) if center_xyz is not None else None
    lib().ora_voxelize_pose(_p(rec_xyz, C.c_float), _p(rec_smt, C.c_int32), len(rec_xyz), _p(lig_xyz, C.c_float),
                            _p(lig_smt, C.c_int32), len(lig_xyz), _p(rmap, C.c_int32), nrc, _p(lmap, C.c_int32),
                            nlc, _p(cin, C.c_float) if cin is not None else None, resolution, dimension,
                            radius_scale, _p(cen, C.c_float), _p(out, C.c_float))
    return out, cen
