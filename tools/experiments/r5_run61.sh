#!/bin/bash
# round 5, GPU call 61: conv3d_h2_16_kernel with its weights requested four steps ahead: Dense parity, B = 1 latency, Dense rate
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_custom_model.py tests/test_gpu_h2_range.py -m gpu -x -q -k "goldens or ensemble or custom or given_grids or batch_independence or range" 2>&1 | tail -3
timeout 600 python tools/experiments/lanes_diag4.py 2>&1 | tail -3
timeout 300 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline %.0f, %.3f ms/step' % (d['value'], d['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense'): print('   ', k, a[k].get('poses_per_s'), a[k].get('blocks_poses_per_s'))
"
