#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy summary of one HIP source as the build compiles it (gfx950).
    python tools/kernel_resources.py gnina_amd/csrc/vina.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnina_amd import build as b  # noqa: E402

src = os.path.abspath(sys.argv[1])
cmd = [b.hipcc()] + b.FLAGS + sys.argv[2:] + ["-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/tmp/_kr.o",
                                               "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|"
                  r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
if not rows:
    sys.exit("no kernels found:\n" + out[-2000:])
try:
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
except OSError:
    names = [r["name"] for r in rows]
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    print(f"{n[:70]:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs','?'):>3s} "
          f"scratch {r.get('ScratchSize','?'):>4s} vspill {r.get('VGPRs Spill','?'):>3s} occ {r.get('Occupancy','?'):>2s} "
          f"lds {r.get('LDS Size','?')}")
