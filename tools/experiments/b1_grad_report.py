"""Kernels of the LAST gradient call in a rocprofv3 --kernel-trace of b1_grad_timeline.py (a call starts with gather_pose_atoms):
python tools/experiments/b1_grad_report.py <trace dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gi = [i for i, r in enumerate(rows) if 'gather_pose_atoms' in r['Kernel_Name']]
n_groups = 2 if any('voxelize_tiles<2' in r['Kernel_Name'] for r in rows) and any('voxelize_tiles<1' in r['Kernel_Name'] for r in rows) else 1
start = gi[-n_groups]
while start > 0 and ('fillBuffer' in rows[start - 1]['Kernel_Name'] or 'copyBuffer' in rows[start - 1]['Kernel_Name']):
    start -= 1
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('q%-3s %-60s start %7.1f dur %6.1f' % (r['Queue_Id'], r['Kernel_Name'].replace('void mig::', '').replace('mig::', '').split('(')[0][:60], (s_ - t0) / 1e3, (e_ - s_) / 1e3))
print('span of the call on the GPU: %.1f us' % ((max(int(r['End_Timestamp']) for r in rows[start:]) - t0) / 1e3))
