"""Registers, spills and scratch of the kernels in the shipped library: python tools/isa/kernel_regs.py [substring ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
try:
    shutil.copy(os.path.join(ROOT, "gnina_amd", "lib", "libmi_gnina.so"), tmp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "libmi_gnina.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in sorted(os.listdir(tmp)):
        if not f.endswith("gfx950"):
            continue
        txt = subprocess.run([f"{LLVM}/llvm-readobj", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
            name = g("name").group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void mig::", "").split("(")[0]
            if sys.argv[1:] and not any(a in dem for a in sys.argv[1:]):
                continue
            agpr = re.match(r"\s*(\d+)", blk).group(1)
            print(f"{dem[:80]:80s} agpr {agpr:>3} vgpr {g('vgpr_count').group(1):>3} sgpr {g('sgpr_count').group(1):>3} "
                  f"spill s{g('sgpr_spill_count').group(1)} v{g('vgpr_spill_count').group(1)} scratch {g('private_segment_fixed_size').group(1)} lds {g('group_segment_fixed_size').group(1)}")
finally:
    shutil.rmtree(tmp)
