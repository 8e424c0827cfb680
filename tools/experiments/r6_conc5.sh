#!/bin/bash
# round 6, call 5: (1) the microbenchmark's victims next to the REAL aggressor; (2) the microbenchmark with an LDS-fed MFMA aggressor;
# (3) what a voxelizer / a library without packed-fp32 instructions costs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== synthetic victims next to a Dense scorer"; timeout 300 python tools/experiments/pk_victim_next_to_scorer.py
echo "== microbenchmark (scalar-load mode + the chain), all aggressor kinds"; PK_SHORT=1 timeout 300 tools/microbench/pk_f32_next_to_mfma | grep -v "waited for"
echo "== bench per library"
for lib in "" voxnopk allnopk; do
  if [ -n "$lib" ]; then export MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_$lib.so; fi
  echo "-- ${lib:-product}"
  timeout 900 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > gpurun_out/r6_bench_${lib:-product}.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6_bench_${lib:-product}.json"))
print("headline", d["value"], "ms/step", d["ms_per_step"], {k["kernel"]: k["ms_per_step"] for k in d["kernels"]})
a=d.get("also",{})
for k in ("crossdock_default2018","dense","fp32_mfma","gradient_calls","seam_b1"):
    if k in a: print(k, json.dumps(a[k])[:400])
PY
done
