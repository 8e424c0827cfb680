#!/usr/bin/env python3
"""LDS bank model of the split-fp16 kernels' A-operand reads (conv3d_h2.hip), with a search over padded strides.

Model as in lds_bank_sim.py: a ds_read_b128 is served in four groups of 16 lanes; a group takes as many LDS cycles as the
largest number of its lanes that hit one 16-byte slot modulo 16 with DIFFERENT addresses (identical addresses broadcast).
The halo tile is [x][y][z][octet][h8 | l8]: slot address of voxel (x, y, z), octet o, half s (0 = h, 1 = l) is
    x * SX + y * SY + z * SZ + 2 * o + s        SZ = 2 * c + pv,  SY = HZ * SZ + py,  SX = HY * SY + px   (16-byte slots)
16-wide kernel: lane = row + 16 * kg, row = cell * 8 + x * 4 + y * 2 + z over two raster-neighbour cells, kg = which of
the step's four octets (channel-major K order: four consecutive (octet, tap) pairs).
32-wide kernel: lane = row + 32 * kh, row as in conv3d.hip (four cells), kh = which of the step's two octets.
Usage: lds_bank_sim_h2.py"""
import itertools

G4 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS = G4 + [[l + 32 for l in g] for g in G4]


def group_cycles(addr):
    tot = 0
    for g in GROUPS:
        per = {}
        for l in g:
            per.setdefault(addr[l] % 16, set()).add(addr[l])
        tot += max(len(v) for v in per.values())
    return tot / 4.0


def sim(tc, c, pv, py, px, wide, stacked=False, halo=1):
    tcx, tcy, tcz = tc
    HY, HZ = 2 * tcy + 2 * halo, 2 * tcz + 2 * halo
    SZ = 2 * c + pv
    SY = HZ * SZ + py
    SX = HY * SY + px
    taps = 27 if halo else 1
    Q = taps * c
    qoff = []
    for q in range(Q + 8):
        qq = min(q, Q - 1)
        c8, tap = divmod(qq, taps)
        dx, dy, dz = (tap // 9, (tap // 3) % 3, tap % 3) if halo else (0, 0, 0)
        qoff.append(dx * SX + dy * SY + dz * SZ + 2 * c8)
    ncell = tcx * tcy * tcz
    tot, n = 0.0, 0
    if wide:  # 32-wide: M-tile = four cells
        for mt in range((ncell + 3) // 4):
            base = []
            for row in range(32):
                oz, oy, ox = row & 1, (row >> 1) & 1, (row >> 3) & 1
                cim = ((row >> 2) & 1) + 2 * ((row >> 4) & 1)
                if stacked:
                    cz, cy, cx = mt % tcz, (mt // tcz) % tcy, 4 * (mt // (tcz * tcy)) + cim
                else:
                    cell = min(mt * 4 + cim, ncell - 1)
                    cz, cy, cx = cell % tcz, (cell // tcz) % tcy, cell // (tcz * tcy)
                base.append((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz) * SZ)
            for pr in range((Q + 1) // 2):
                addr = [base[l & 31] + qoff[2 * pr + (l >> 5)] for l in range(64)]
                tot, n = tot + group_cycles(addr), n + 1
    else:
        for mt in range((ncell + 1) // 2):
            base = []
            for row in range(16):
                oz, oy, ox, cim = row & 1, (row >> 1) & 1, (row >> 2) & 1, row >> 3
                cell = min(mt * 2 + cim, ncell - 1)
                cz, cy, cx = cell % tcz, (cell // tcz) % tcy, cell // (tcz * tcy)
                base.append((2 * cx + ox) * SX + (2 * cy + oy) * SY + (2 * cz + oz) * SZ)
            for st in range((Q + 3) // 4):
                addr = [base[l & 15] + qoff[4 * st + (l >> 4)] for l in range(64)]
                tot, n = tot + group_cycles(addr), n + 1
    return tot / n, SX * (2 * tcx + 2 * halo) * 16


def search(tc, c, wide, stacked=False, halo=1):
    best = None
    for pv, py, px in itertools.product((1, 3, 5, 7), range(0, 16), range(0, 16)):
        m, size = sim(tc, c, pv, py, px, wide, stacked, halo)
        key = (round(m, 3), size)
        if best is None or key < best[0]:
            best = (key, (pv, py, px))
    return best


if __name__ == "__main__":
    print("kernel tile(cells) c  layout      now: mean LDS cycles/group (1.00 = conflict free)   best pads (pv, py, px): mean, bytes")
    for wide, tc, c, stacked, halo in ((False, (2, 4, 4), 2, False, 1), (False, (2, 4, 4), 1, False, 1), (False, (2, 2, 6), 2, False, 1),
                                       (False, (3, 3, 3), 2, False, 1), (True, (4, 4, 2), 2, True, 1), (True, (4, 4, 2), 1, True, 1),
                                       (True, (2, 2, 6), 2, False, 1), (True, (3, 3, 3), 2, False, 1), (True, (2, 2, 4), 6, False, 0)):
        now, size = sim(tc, c, 1, 0, 0, wide, stacked, halo)
        b = search(tc, c, wide, stacked, halo)
        print(f"{'32-wide' if wide else '16-wide'} {str(tc):10s} {c}  {'stacked-x' if stacked else 'raster   '}  {now:5.2f} ({size} B)   ->  {b[1]}: {b[0][0]:.2f}, {b[0][1]} B")
