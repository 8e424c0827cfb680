#!/bin/bash
# round 5, GPU call 4: the whole -m gpu suite on the current tree (lean push: the 57 build-time weight blobs stay behind,
# their per-model tests skip), k1s probes, C5 entry
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'conv1' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for dbg in 0 2 4 6 64 70; do
  echo "== dense, MI_GNINA_K1S_DBG=$dbg"
  MI_GNINA_K1S_DBG=$dbg timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
echo "== C5"
timeout 900 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from gnina_amd import capi, synth
capi.init(0)
print(json.dumps(bench.config_c5(capi, synth), indent=1, default=float))
PY
