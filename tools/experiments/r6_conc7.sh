#!/bin/bash
# round 6, call 7: (1) the stand-alone program with a k1s-like aggressor (f16 MFMA fed by global loads); (2) the product
# build (no packed fp32 anywhere, no per-device lock): the reproducers, the concurrency tests, the whole GPU suite, the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== microbenchmark: the chain next to every aggressor kind"; PK_SHORT=1 timeout 300 tools/microbench/pk_f32_next_to_mfma
echo "== product: voxelizer alone next to dense_1_3"
timeout 300 python tools/experiments/vox_stress.py --iters 50000 --label "product" | grep -v "^  iteration"
echo "== product: two scorers on two threads"
DIAG_CALLS=2500 timeout 900 python tools/experiments/concurrency_diag.py dense_1_3,crossdock_default2018_KD_4 dense_1_3,dense_1_3_PT_KD_3 2>&1 | tail -4
echo "== concurrency tests"; timeout 900 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -5
echo "== whole GPU suite"; timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== bench"; timeout 1500 python bench.py 2>gpurun_out/r6_bench7.err | tail -1 > gpurun_out/r6_bench7.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_bench7.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"])
a=d["also"]
print("seam_b1", json.dumps(a.get("seam_b1"))[:1500])
print("dense", json.dumps(a.get("dense"))[:300])
print("gradient_calls", json.dumps(a.get("gradient_calls"))[:600])
PY
