#!/bin/bash
# round 5, GPU call 51: the headline with the clocks spun up before the warm-up steps (bench.py spin_up), against --spinup-seconds 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('   headline %.0f %s, %.3f ms/step, spinup %s, roofline %.3f (%.3f ms)' % (d['value'], d['unit'], d['ms_per_step'], d['spinup']['steps'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))
a=d.get('also',{})
for k in ('crossdock_default2018','dense'): print('   ', k, a.get(k,{}).get('poses_per_s'), a.get(k,{}).get('blocks_poses_per_s'))
"; }
for sp in 0.3 0.3 0.3; do echo "== spinup $sp"; timeout 300 python bench.py --no-configs --no-cpu-baseline --spinup-seconds $sp 2>/dev/null | kern; done
