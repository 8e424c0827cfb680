#!/bin/bash
# round 5, GPU call 6 (second session): the state of the tree -- whole -m gpu suite, the full bench line, kernel stats + PMC
# passes of the headline and of the Dense model
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench (full)"
timeout 900 python bench.py > gpurun_out/r5/bench_full.json 2> gpurun_out/r5/bench_full.err; tail -c 400 gpurun_out/r5/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_full.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
print(json.dumps(d.get('also'), default=float)[:6000])
print(json.dumps(d.get('cpu_baseline'), default=float))
PY
echo "== dense"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null > gpurun_out/r5/bench_dense.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_dense.json').read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
PY
echo "== profiles"
bash tools/profile_gpu.sh r5a > gpurun_out/r5/prof_r5a.log 2>&1
bash tools/profile_gpu.sh r5a_dense --model dense > gpurun_out/r5/prof_r5a_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_r5a > gpurun_out/r5/pmc_r5a.txt
python tools/pmc_summary.py gpurun_out/prof_r5a_dense > gpurun_out/r5/pmc_r5a_dense.txt
ls gpurun_out/prof_r5a/trace
