#!/bin/bash
# round 5, GPU call 5: persistent d16 / k1s launches, lanes for small ensemble calls, tests touched by the Dense work
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
echo "== pytest (dense split, refine, gradient, h2, parity, host adapter)"
timeout 1500 python -m pytest tests/test_gpu_dense_split.py tests/test_gpu_cnn_refine.py tests/test_gpu_gradient.py tests/test_gpu_h2.py tests/test_gpu_parity.py tests/test_host_adapter.py tests/test_gpu_h2_range.py -m gpu -q 2>&1 | tail -15
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
blk=sum(k['ms_per_step'] for k in d.get('kernels', []) if 's24' in k['kernel'] and 'to16_sp' in k['kernel'])
blk1=sum(k['ms_per_step'] for k in d.get('kernels', []) if 's12' in k['kernel'] and 'to16_sp' in k['kernel'])
print('   block0 (24^3 d16 layers) %.3f ms   block1 (12^3) %.3f ms' % (blk, blk1))
for k in d.get('kernels', []):
    if 'conv1' in k['kernel'] or '28to32' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
echo "== dense (persistent, defaults)"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== dense MI_GNINA_D16_PERSIST=0 MI_GNINA_K1S_PERSIST=0"
MI_GNINA_D16_PERSIST=0 MI_GNINA_K1S_PERSIST=0 timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== dense MI_GNINA_D16_PERSIST=3"
MI_GNINA_D16_PERSIST=3 timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== dense MI_GNINA_D16_NP=1"
MI_GNINA_D16_NP=1 timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | kern
echo "== seam B=1 (lanes) / MI_GNINA_NO_LANES=1"
for nl in "" 1; do
MI_GNINA_NO_LANES=$nl timeout 600 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from gnina_amd import capi, synth
capi.init(0)
print(os.environ.get("MI_GNINA_NO_LANES"), json.dumps(bench.config_seam_b1(capi, synth), default=float))
PY
done
