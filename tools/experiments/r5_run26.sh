#!/bin/bash
# persistent voxelizer wavefronts: parity, then A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid96.py tests/test_gpu_h2.py tests/test_gpu_gradient.py -m gpu -x -q 2>&1 | tail -5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'vox' in k['kernel'] or 'gather' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for env in "MI_VOX_PERSIST=0" "" "MI_VOX_PERSIST=16" "MI_VOX_PERSIST=24" "MI_VOX_PERSIST=64" "MI_VOX_PERSIST=128" "MI_VOX_DBG=7" "MI_VOX_DBG=8"; do
  echo "== default2017 [$env]"
  env $env timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
echo "== dense_1_3 @ 96^3 forward (C5 grid): voxelizer share"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
