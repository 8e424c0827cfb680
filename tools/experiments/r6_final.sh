#!/bin/bash
# round 6, final call: whole GPU suite, smoke, the bench line, kernel stats + PMC passes of the headline and of the Dense model
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r6
echo "== GPU suite"; timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench (full)"
timeout 1500 python bench.py > gpurun_out/r6/bench_final.json 2> gpurun_out/r6/bench_final.err; tail -c 300 gpurun_out/r6/bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_final.json').read().strip().splitlines()[-1])
print('poses/s %.0f (without spin-up %.0f)  ms/step %.3f' % (d['value'], d.get('value_without_spinup', 0), d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
a=d['also']
for k in ('crossdock_default2018','dense','fp32_mfma_only'): print(k, a.get(k, {}).get('poses_per_s'))
print('c3', a['c3'].get('mc_s'), a['c3'].get('strict_mode'))
print('c3_real', a['c3_real'].get('mc_s'), a['c3_real'].get('strict_order_mode'), a['c3_real'].get('gpu_over_reference_cpu'), a['c3_real'].get('chains_bit_identical_to_reference'))
print('c4', a['c4'].get('ligands_per_s'))
print('c5', {k: (v.get('poses_per_s_forward'), v.get('poses_per_s_forward_backward')) for k, v in a['c5'].items() if isinstance(v, dict) and 'poses_per_s_forward' in v}, a['c5'].get('refine'))
print('seam', json.dumps({k: a['seam_b1'][k] for k in ('default2017', 'default_ensemble')}))
print('grad', json.dumps({k: v for k, v in a['gradient_calls'].items() if k != 'note'}, default=float)[:900])
print(json.dumps(d.get('cpu_baseline'), default=float)[:400])
PY
echo "== profiles"
bash tools/profile_gpu.sh r6g > gpurun_out/r6/prof_r6g.log 2>&1
bash tools/profile_gpu.sh r6g_dense --model dense > gpurun_out/r6/prof_r6g_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r6g > gpurun_out/r6/pmc_r6g.txt
python tools/pmc_summary.py gpurun_out/prof_r6g_dense > gpurun_out/r6/pmc_r6g_dense.txt
python tools/pmc_summary.py gpurun_out/prof_r6g --json gpurun_out/r6/pmc_r6g.json 2>/dev/null | tail -1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_r6g/trace_default -o t -- python $R/bench.py --no-cpu-baseline --no-configs --extras-timeout 1 > $R/gpurun_out/r6/trace_default.log 2>&1
ls $R/gpurun_out/prof_r6g/trace $R/gpurun_out/prof_r6g/trace_default 2>/dev/null | head
