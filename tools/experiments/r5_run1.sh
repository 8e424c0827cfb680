#!/bin/bash
# round 5, GPU call 1: Dense on split tensors (first run) + the voxelizer's conflict-free transpose
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
kern() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense split tests"
timeout 900 python -m pytest tests/test_gpu_dense_split.py -x -q -s 2>&1 | tail -40
echo "== voxelizer parity"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_h2.py -x -q 2>&1 | tail -8
echo "== bench dense (split)"
timeout 600 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== bench dense (round-4 kernels: MI_GNINA_NO_DENSE_SPLIT)"
MI_GNINA_NO_DENSE_SPLIT=1 timeout 600 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== bench dense NP=1"
MI_GNINA_D16_NP=1 timeout 600 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== bench default2017"
timeout 600 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
