#!/bin/bash
# GPU box: per-kernel times of the Dense family under experiment switches (one bench run per argument)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() {
  echo "== $*"
  env "$@" python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    print('   %-34s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"
}
if [ $# -eq 0 ]; then set -- A=0 MI_GNINA_H2_DBG=8; fi
for cfg in "$@"; do run $(echo $cfg | tr ',' ' '); done
