#!/bin/bash
# round 5, GPU call 65: the bench line after the last edits of bench.py (short form)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 120 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline %.0f, %.3f ms/step, steps %d warmup %d' % (d['value'], d['ms_per_step'], d['steps'], d['warmup']))
print(json.dumps(d['voxelizer'])[:700])
print(json.dumps(d['roofline'])[:300])
"
