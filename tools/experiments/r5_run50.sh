#!/bin/bash
# round 5, GPU call 50: where the headline step's time goes between its kernels (kernel trace with timestamps of the timed loop)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/prof_r5gap; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > $OUT/bench.log 2>&1
tail -c 400 $OUT/bench.log | head -c 300; echo
cd $R; python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r5gap/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the headline's timed loop: consecutive runs of gather -> voxelize -> conv x3 -> fc
names = [r['Kernel_Name'] for r in rows]
def short(n): return n.replace('void mig::','').replace('mig::','').split('(')[0][:40]
seq = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# find steps: index of gather_pose_atoms followed by voxelize_tiles<1, true>
steps = [i for i in range(len(seq)-6) if seq[i][0].startswith('gather_pose_atoms') and seq[i+1][0].startswith('voxelize_tiles<1, true>')]
print('steps found', len(steps))
import statistics
# take the steps 3..12 (after warmup): print per-kernel durations and the gap before each kernel
for k in range(4, min(len(steps)-1, 9)):
    i = steps[k]; j = steps[k+1]
    line = []
    for q in range(i, j):
        n, s, e = seq[q]
        gap = s - seq[q-1][2]
        line.append('%s gap %.1f dur %.1f' % (n[:22], gap/1e3, (e-s)/1e3))
    print('step', k, 'total %.1f us' % ((seq[j][1]-seq[i][1])/1e3))
    for l in line: print('    ', l)
PY
