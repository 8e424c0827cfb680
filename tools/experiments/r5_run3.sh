#!/bin/bash
# round 5, GPU call 3: cheap probes -- poses per internal chunk (MALL residency of the activations), the d16 epilogue,
# tiles of the 1x1x1 transitions (z-run length of the DMA sources)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
blk=sum(k['ms_per_step'] for k in d.get('kernels', []) if 's24' in k['kernel'] and 'to16_sp' in k['kernel'])
print('   block0 (24^3 d16 layers) %.3f ms' % blk)
for k in d.get('kernels', []):
    if 'conv1' in k['kernel'] or '28to32' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for ch in 32 64 128 256 512; do
  echo "== dense --chunk $ch"
  timeout 300 python bench.py --model dense --chunk $ch --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
for dbg in 64 78; do
  echo "== dense, MI_GNINA_H2_DBG=$dbg"
  MI_GNINA_H2_DBG=$dbg timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
for t in 1,1,12 1,2,6 2,1,6 1,4,4 2,2,3 4,2,2; do
  echo "== dense, MI_GNINA_K1_TILE=$t"
  MI_GNINA_K1_TILE=$t timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
