# usage: run_env.sh "ENV=1 [ENV2=..]" ...   -- one bench.py summary line per environment setting (BENCH_ARGS adds flags)
for e in "$@"; do
  env $e python bench.py --no-configs --no-cpu-baseline --steps 20 $BENCH_ARGS > gpurun_out/env.json 2>gpurun_out/env.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/env.json").read().strip().splitlines()[-1])
print("$e", d["value"], d["ms_per_step"], [(k["kernel"][:22],k["ms_per_step"]) for k in d["kernels"]])
PY
done
