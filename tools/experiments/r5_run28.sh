#!/bin/bash
# k1s without its spills (opaque thread index for the per-item set-up and epilogue)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -4
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f  sum of kernels %.3f' % (d['value'], d['ms_per_step'], d.get('sum_kernel_ms_per_step', 0)))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
