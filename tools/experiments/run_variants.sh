for v in "$@"; do
  if [ $v = base ]; then unset MI_GNINA_LIB; else export MI_GNINA_LIB=$PWD/gnina_amd/lib/variants/libmi_gnina_$v.so; fi
  python bench.py --no-configs --no-cpu-baseline --steps 20 > gpurun_out/var_$v.json 2>gpurun_out/var_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/var_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["dense_no_skip"]["avg_launch_ms"])
PY
done
