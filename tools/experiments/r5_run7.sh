#!/bin/bash
# round 5, GPU call 7: (1) which part of a small ensemble call misses the goldens (lanes?), (2) persistent conv3d_h2_kernel
# (first convolutions): parity tests, then A/B timings -- MI_GNINA_H2_PERSIST=0 (one workgroup per item, the round-4 kernel),
# default (persistent + early DMA), MI_GNINA_H2_EARLY=0, a cap of 2 workgroups per CU
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
echo "== lanes diagnostic"
timeout 600 python tools/experiments/lanes_diag.py 2>&1 | tail -20
echo "== parity tests on the persistent kernel"
timeout 1200 python -m pytest tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_gpu_dense_split.py tests/test_gpu_h2_range.py -m gpu -x -q 2>&1 | tail -8
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'conv3_s24' in k['kernel'] or 'vox' in k['kernel'] or 's12_32' in k['kernel'] or 's6_64' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for env in "" "MI_GNINA_H2_PERSIST=0" "MI_GNINA_H2_EARLY=0" "MI_GNINA_H2_PERSIST=2" "MI_GNINA_H2_PERSIST=0 MI_GNINA_H2_DBG=2" "MI_GNINA_H2_DBG=2" "MI_GNINA_H2_DBG=6" "MI_GNINA_H2_PERSIST=0 MI_GNINA_H2_DBG=6"; do
  echo "== default2017 [$env]"
  env $env timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
for m in crossdock_default2018 dense; do
  for env in "" "MI_GNINA_H2_PERSIST=0"; do
    echo "== $m [$env]"
    env $env timeout 300 python bench.py --model $m --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
  done
done
