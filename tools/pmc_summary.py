#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 counter_collection CSVs (one or more --pmc passes).
Usage: pmc_summary.py <prof_dir>   (expects <prof_dir>/pmc_*/p_counter_collection.csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void mig::", "").replace("mig::", "")
    return n.split("(")[0][:48]


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    print(f"# per-dispatch averages from {d}")
    for k, cs in acc.items():
        print(k)
        for c in counters:
            if c in cs:
                v = cs[c]
                print(f"    {c:32s} {sum(v) / len(v):18.1f}   (n={len(v)})")


def to_json(d, out):
    """Write per-kernel per-dispatch averages as JSON (bench.py reads profiles/latest_pmc.json for
    roofline.traffic).  HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE
    are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads, so the
    read side is doubled; WRITE_SIZE is taken as is (it matches the known pooled-output bytes here)."""
    import json
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        if e.get("SQ_INSTS_VALU_MFMA_MOPS_F32"):
            # one MOP = 512 FLOP (a 32x32x2 fp32 MFMA = 4096 FLOP = 8 MOPs)
            e["mfma_executed_flops_per_launch"] = e["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
        if e.get("SQ_INSTS_VALU_MFMA_MOPS_F16"):
            # split-fp16 kernels: a 32x32x16 f16 MFMA = 32,768 FLOP = 64 MOPs
            e["mfma_executed_flops_per_launch"] = e.get("mfma_executed_flops_per_launch", 0.0) + e["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512.0
        res[k] = e
    # bench.py's label of the dominant kernel -> profiler kernel name (first conv of the default workload)
    kmap = {}
    first = sorted((k for k in res if k.startswith("conv3d_mfma_kernel<4, 1, 2, 1")), key=lambda k: "1, true" not in k)  # the zero-skipping instantiation first
    if first:
        kmap["conv3_s24_35to32_pool"] = first[0]
        kmap["conv3_s24_28to32"] = first[0]
    h2 = sorted((k for k in res if k.startswith("conv3d_h2_kernel<4, 1, 2, 1")), key=lambda k: "true, true" not in k)
    if h2:  # the default path's first conv (zero-skipping, x-stacked M-tiles)
        kmap["conv3_s24_35to32_pool_h2"] = h2[0]
        kmap["conv3_s24_28to32_h2"] = h2[0]
    json.dump({"source": d, "command": "tools/profile_gpu.sh (bench.py --steps 3 --warmup 1 --no-cpu-baseline)",
               "roofline_kernel_map": kmap, "kernels": res}, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--json":
        to_json(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1])
