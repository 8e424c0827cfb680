#!/bin/bash
# round 6, call 3: the stand-alone program (packed fp32 next to another kernel's MFMA waves), the voxelizer / the whole library
# compiled without packed-fp32 instructions under the round-5 reproducer with the per-device lock off, and what that costs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== microbenchmark"; timeout 300 tools/microbench/pk_f32_next_to_mfma
export MI_GNINA_NO_CALL_LOCK=1
for lib in voxnopk allnopk; do
  export MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_$lib.so
  echo "== $lib: voxelizer alone next to dense_1_3"
  timeout 200 python tools/experiments/vox_stress.py --iters 50000 --label "$lib"
  echo "== $lib: two scorers on two threads, no lock"
  DIAG_CALLS=1500 timeout 600 python tools/experiments/concurrency_diag.py dense_1_3,crossdock_default2018_KD_4 dense_1_3,dense_1_3_PT_KD_3 dense_1_3,default2017 2>&1 | tail -6
done
unset MI_GNINA_LIB MI_GNINA_NO_CALL_LOCK
echo "== headline per library"
for lib in "" voxnopk allnopk; do
  if [ -n "$lib" ]; then export MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_$lib.so; fi
  echo "-- ${lib:-product}"
  timeout 600 python bench.py --no-cpu-baseline --no-configs --extras-timeout 1 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v for k,v in d.get('kernels',{}).items()} if 'kernels' in d else d.get('roofline'))"
done
