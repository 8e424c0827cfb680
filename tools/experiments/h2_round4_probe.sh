#!/bin/bash
# GPU box: the headline kernels' times under the split-fp16 kernels' experiment switches (gpurun_out/r04_probe.txt)
# usage: h2_round4_probe.sh [ENV=VAL,ENV=VAL ...]   (one bench run per argument; "A=0" = defaults)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {
  echo "== $*"
  env "$@" python bench.py --no-configs --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    print('   %-32s %.4f ms' % (k['kernel'], k['ms_per_step']))
for n in ('crossdock_default2018', 'dense'):
    if n in d.get('also', {}): print('   also %-28s %.0f poses/s' % (n, d['also'][n]['poses_per_s']))
"
}
if [ $# -eq 0 ]; then set -- A=0 MI_GNINA_H2_NO_SPLIT_TENSORS=1 MI_GNINA_H2_DBG=2 MI_GNINA_H2_DBG=6; fi
for cfg in "$@"; do run $(echo $cfg | tr ',' ' '); done
