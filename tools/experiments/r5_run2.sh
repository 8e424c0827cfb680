#!/bin/bash
# round 5, GPU call 2: what bounds conv3d_h2_d16_kernel -- timing-only switches (MI_GNINA_H2_DBG bits: 2 no K loop, 4 no
# tile DMA, 8 no weight DMA, 32 contiguous tile sources) and cache counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
mkdir -p gpurun_out/r5
kern() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('poses/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))
for k in d.get('kernels', []):
    if 'sp_h2' in k['kernel'] or 'vox' in k['kernel'] or '28to32' in k['kernel']: print('   %-40s x%-2d %.4f ms' % (k['kernel'], k['launches_per_step'], k['ms_per_step']))
"; }
for dbg in 0 2 4 8 32 34 12 14; do
  echo "== dense, MI_GNINA_H2_DBG=$dbg"
  MI_GNINA_H2_DBG=$dbg timeout 300 python bench.py --model dense --no-configs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | kern
done
export TMPDIR=/tmp
OUT=$R/gpurun_out/r5/pmc_dense; mkdir -p $OUT; cd /tmp
BENCH="python $R/bench.py --model dense --steps 2 --warmup 1 --no-cpu-baseline --no-configs"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -f csv -d $OUT/a -o p -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -f csv -d $OUT/b -o p -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/c -o p -- $BENCH > $OUT/c.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUSY_avr --kernel-trace -f csv -d $OUT/d -o p -- $BENCH > $OUT/d.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0]
        if 'd16' in k or 'k1s' in k or 'conv3d_h2_kernel' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-34s mean %.4g  n %d  min %.4g max %.4g' % (c, sum(x)/len(x), len(x), min(x), max(x)))
PY
tail -2 $OUT/a.log $OUT/d.log | cut -c1-300
