#!/bin/bash
# round 6, call 2: the deviating lanes are the wave's upper half during the accumulation (lz = 2, 3 of a sub-block: cells with odd
# cz, all (cx, cy) of the sub-block), values of e^-2 (2 d/r - 3)^2 BEYOND the support -- lanes that executed density_add's
# body though their own compare was false.  Which instruction pair is short of wait states next to MFMA waves?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export MI_GNINA_NO_CALL_LOCK=1
V="timeout 200 python tools/experiments/vox_stress.py --iters 20000"
$V --label "product build"
for f in fix1 fix2 fix3; do
  MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_$f.so $V --label "$f"
done
MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_trap.so $V --flags 8 --label "trap (exec-violation check)"
MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_trap.so $V --flags 8 --no-aggressor --label "trap, no aggressor"
