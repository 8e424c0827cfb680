#!/bin/bash
# round 5, GPU calls 43, 44: the voxelizer with a straight-line density_add (44: + timing switches compiled out, SGPR-offset s_loads) (the kernel issues more SALU than VALU instructions:
# 543 M against 468 M per launch -- the nested zone branches' exec-mask bookkeeping): bits (voxel parity tests), then its time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "voxelize or typer or goldens or nearly_empty" 2>&1 | tail -5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('   headline %.0f %s, %.3f ms/step' % (d['value'], d['unit'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'vox' in k['kernel'] or 'gather' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for i in 1 2; do timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | kern; done
