#!/usr/bin/env python3
"""Convert the reference's gninagrid golden grids into one compact fixture.

Run in the build container (needs /root/reference):
    python tests/golden/make_voxel_goldens.py
Reads   /root/reference/test/gninagrid/files/{ccdx,ccmap,ccbin}_0_{rec,lig}_AliphaticCarbonXSHydrophobe.*,
        ccgrid_0.25.29.binmap, usergrid.dx, CC.xyz, C.xyz, recmap, ligmap
        (produced by the reference's own `gninagrid` runs, test/gninagrid/CMakeLists.txt:18-34)
Writes  tests/golden/voxel_goldens.npz  (values only -- no reference source is copied).
The fixture travels to the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np

REF = "/root/reference/test/gninagrid/files/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voxel_goldens.npz")


def read_dx(path):
    lines = open(path).read().split("\n")
    n = tuple(int(x) for x in lines[0].split()[-3:])
    origin = np.array([float(x) for x in lines[1].split()[1:]], dtype=np.float64)
    vals = np.array(" ".join(lines[7:]).split(), dtype=np.float64)
    return origin, vals.reshape(n).astype(np.float32)  # dx order: x slowest, z fastest


def read_map(path):
    lines = open(path).read().split("\n")
    hdr = {l.split()[0]: l.split()[1:] for l in lines[:6] if l.strip()}
    npts = [int(x) + 1 for x in hdr["NELEMENTS"]]
    vals = np.array([float(x) for x in lines[6:] if x.strip()], dtype=np.float32)
    # AutoDock .map order is x fastest -> transpose to [x][y][z]
    grid = vals.reshape(npts[2], npts[1], npts[0]).transpose(2, 1, 0).copy()
    return np.array([float(x) for x in hdr["CENTER"]]), float(hdr["SPACING"][0]), grid


def read_xyz_heavy(path):
    rows = [l.split() for l in open(path).read().split("\n")[2:] if l.strip()]
    # OpenBabel ingest deletes non-polar hydrogens (GninaConverter.cpp:100): keep heavy atoms
    return np.array([[float(v) for v in r[1:4]] for r in rows if r[0] != "H"], dtype=np.float32)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference goldens not available: " + REF)
    out = {}
    out["recmap"] = np.array(open(REF + "recmap").read())
    out["ligmap"] = np.array(open(REF + "ligmap").read())
    out["cc_xyz"] = read_xyz_heavy(REF + "CC.xyz")
    out["c_xyz"] = read_xyz_heavy(REF + "C.xyz")
    for side in ("rec", "lig"):
        o, g = read_dx(REF + f"ccdx_0_{side}_AliphaticCarbonXSHydrophobe.dx")
        out[f"ccdx_{side}_origin"], out[f"ccdx_{side}"] = o, g
        c, sp, g = read_map(REF + f"ccmap_0_{side}_AliphaticCarbonXSHydrophobe.map")
        out[f"ccmap_{side}_center"], out[f"ccmap_{side}"] = c, g
        o, g = read_dx(REF + f"ccbin_0_{side}_AliphaticCarbonXSHydrophobe.dx")
        out[f"ccbin_{side}_origin"], out[f"ccbin_{side}"] = o, g
    o, g = read_dx(REF + "usergrid.dx")
    out["usergrid_origin"], out["usergrid"] = o, g
    out["ccgrid_binmap"] = np.fromfile(REF + "ccgrid_0.25.29.binmap", dtype=np.float32).reshape(29, 25, 25, 25)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
