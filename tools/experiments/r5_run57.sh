#!/bin/bash
# round 5, GPU call 57: the timeline of one B = 1 call of gnina's default ensemble with lanes (kernel trace, per queue)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/prof_r5b1; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python $R/tools/experiments/b1_timeline.py > $OUT/log.txt 2>&1
grep "median call" $OUT/log.txt
cd $R; python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r5b1/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('void mig::','').replace('mig::','').split('(')[0][:34]
# the last call: from the last-but-one gather pair to the end
gi = [i for i, r in enumerate(rows) if 'gather_pose_atoms' in r['Kernel_Name']]
start = gi[-2]   # two groups per call: the last call starts at the second-to-last gather
t0 = int(rows[start]['Start_Timestamp'])
prev_end = {}
for r in rows[start:]:
    q = r['Queue_Id']; s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('q%-3s %-34s start %7.1f dur %6.1f' % (q, short(r['Kernel_Name']), (s_ - t0) / 1e3, (e_ - s_) / 1e3))
print('span of the call on the GPU: %.1f us' % ((max(int(r['End_Timestamp']) for r in rows[start:]) - t0) / 1e3))
PY
