#!/bin/bash
# round 6: bench.py's seam numbers (per-pose calls, 1 / 2 / 4 threads) and Dense / headline throughput after the per-pose work
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python bench.py --only seam_b1 > gpurun_out/r6_b1i_bench.json 2> gpurun_out/r6_b1i_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6_b1i_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['also']['seam_b1'], indent=1)[:3000])
print({k: v.get('poses_per_s') for k, v in d['also'].get('other_models', {}).items()} if 'other_models' in d['also'] else list(d['also'].keys()))
PY
