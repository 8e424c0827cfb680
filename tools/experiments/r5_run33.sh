#!/bin/bash
# round 5, final profiles: the full bench line, kernel stats + PMC passes of the headline and of the Dense model
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
echo "== bench (full)"
timeout 900 python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; tail -c 300 gpurun_out/r5/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_final.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense','fp32_mfma_only'): print(k, a[k].get('poses_per_s'))
print('c3', a['c3'].get('mc_s'), a['c3'].get('strict_mode'))
print('c3_real', a['c3_real'].get('mc_s'), a['c3_real'].get('chains_bit_identical_to_reference'), a['c3_real']['cpu_baseline'].get('value'))
print('c4', a['c4'].get('ligands_per_s'))
print('c5', json.dumps(a['c5'], default=float)[:1500])
print('seam', json.dumps(a['seam_b1'], default=float)[-700:])
print('grad', json.dumps(a['gradient_calls'], default=float)[-900:])
print(json.dumps(d.get('cpu_baseline'), default=float))
PY
echo "== profiles"
bash tools/profile_gpu.sh r5f > gpurun_out/r5/prof_r5f.log 2>&1
bash tools/profile_gpu.sh r5f_dense --model dense > gpurun_out/r5/prof_r5f_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r5f > gpurun_out/r5/pmc_r5f.txt
python tools/pmc_summary.py gpurun_out/prof_r5f_dense > gpurun_out/r5/pmc_r5f_dense.txt
ls gpurun_out/prof_r5f/trace
