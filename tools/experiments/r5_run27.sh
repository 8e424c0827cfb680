#!/bin/bash
# poses per internal chunk: does the pooled grid (2.2 MB per pose) staying in the 256 MB MALL between the voxelizer and the first conv pay?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f  sum of kernels %.3f' % (d['value'], d['ms_per_step'], d.get('sum_kernel_ms_per_step', 0)))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for ch in 0 512 256 128 96 64 32; do
  echo "== default2017 --chunk $ch"
  timeout 300 python bench.py --chunk $ch --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
for ch in 0 256 128 64; do
  echo "== crossdock_default2018 --chunk $ch"
  timeout 300 python bench.py --model crossdock_default2018 --chunk $ch --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern | head -1
done
