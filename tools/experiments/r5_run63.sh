#!/bin/bash
# round 5, GPU call 63: final bench line + kernel stats + PMC passes of the headline and of the Dense model
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
echo "== bench (full)"
timeout 900 python bench.py > gpurun_out/r5/bench_final2.json 2> gpurun_out/r5/bench_final2.err; tail -c 300 gpurun_out/r5/bench_final2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_final2.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'), d.get('spinup'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense','fp32_mfma_only'): print(k, a[k].get('poses_per_s'))
print('c3', a['c3'].get('mc_s'), a['c3'].get('strict_mode'))
print('c3_real', a['c3_real'].get('mc_s'), a['c3_real'].get('chains_bit_identical_to_reference'))
print('c4', a['c4'].get('ligands_per_s'))
print('c5', {k: (v.get('poses_per_s_forward'), v.get('poses_per_s_forward_backward')) for k, v in a['c5'].items() if isinstance(v, dict) and 'poses_per_s_forward' in v}, a['c5'].get('refine'))
print('seam', json.dumps({k: a['seam_b1'][k] for k in ('default2017', 'default_ensemble')}))
print('grad', json.dumps({k: v for k, v in a['gradient_calls'].items() if k != 'note'}, default=float))
print(json.dumps(d.get('cpu_baseline'), default=float)[:400])
PY
echo "== profiles"
bash tools/profile_gpu.sh r5g > gpurun_out/r5/prof_r5g.log 2>&1
bash tools/profile_gpu.sh r5g_dense --model dense > gpurun_out/r5/prof_r5g_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r5g > gpurun_out/r5/pmc_r5g.txt
python tools/pmc_summary.py gpurun_out/prof_r5g_dense > gpurun_out/r5/pmc_r5g_dense.txt
ls gpurun_out/prof_r5g/trace
