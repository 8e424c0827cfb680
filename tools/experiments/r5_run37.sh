#!/bin/bash
# planar split format ([octet][h | l][voxel][8 fp16]): the whole GPU suite, then the bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
kern() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('bench failed', e); sys.exit(0)
print('poses/s %.0f  ms/step %.3f  sum of kernels %.3f' % (d['value'], d['ms_per_step'], d.get('sum_kernel_ms_per_step', 0)))
for k in d.get('kernels', []): print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for rep in 1 2; do
echo "== default2017 ($rep)"
timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern
done
echo "== crossdock"
timeout 300 python bench.py --model crossdock_default2018 --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern | head -5
echo "== dense"
timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | kern
for dbg in 2 6; do
echo "== default2017 MI_GNINA_H2_DBG=$dbg"
MI_GNINA_H2_DBG=$dbg timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | kern | sed -n 4p
done
