#!/bin/bash
# GPU box: instruction counters of the voxelizer in the bench command (gpurun_out/vox_pmc_<tag>.txt)
TAG=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/vox_pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-trace -f csv -d $OUT/a -o p -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC --kernel-trace -f csv -d $OUT/b -o p -- $BENCH > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if 'voxelize_tiles' in r['Kernel_Name']: acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-28s %.4g' % (c, sum(x)/len(x)))
PY
