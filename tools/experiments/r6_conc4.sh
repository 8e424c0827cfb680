#!/bin/bash
# round 6, call 4: narrowing the packed-fp32 fault -- SGPR-pair operands? a scalar load in flight?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== microbenchmark, scalar-load modes"; PK_SHORT=1 timeout 300 tools/microbench/pk_f32_next_to_mfma
export MI_GNINA_NO_CALL_LOCK=1
for f in fix4 fix5; do
  MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_$f.so timeout 200 python tools/experiments/vox_stress.py --iters 30000 --label "$f" | grep -v "^  iteration"
done
