#!/bin/bash
# d16: up to four octets per K chunk for small calls
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_concurrency.py tests/test_host_adapter.py tests/test_gpu_custom_model.py -m gpu -x -q 2>&1 | tail -4
for cc in 1 2 4; do
  echo "== B = 1, MI_GNINA_D16_CC=$cc"
  MI_GNINA_D16_CC=$cc timeout 300 python tools/experiments/seam_b1_breakdown.py dense_1_3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-1300
done
echo "== default (auto)"
timeout 300 python tools/experiments/seam_b1_breakdown.py dense_1_3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
kern() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'sp_h2' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense throughput"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
