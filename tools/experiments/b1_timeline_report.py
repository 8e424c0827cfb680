"""Per-queue timeline of the LAST call in a rocprofv3 --kernel-trace of tools/experiments/b1_timeline.py:
python tools/experiments/b1_timeline_report.py <trace dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    return n.replace('void mig::', '').replace('mig::', '').split('(')[0][:44]


# a call ends with ensemble_reduce_kernel (+ copies): the last call starts behind the last-but-one reduce's copies
ri = [i for i, r in enumerate(rows) if 'ensemble_reduce' in r['Kernel_Name']]
start = ri[-2] + 1
while 'copyBuffer' in rows[start]['Kernel_Name']:
    start += 1
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('q%-3s %-44s start %7.1f dur %6.1f' % (r['Queue_Id'], short(r['Kernel_Name']), (s_ - t0) / 1e3, (e_ - s_) / 1e3))
print('span of the call on the GPU: %.1f us' % ((max(int(r['End_Timestamp']) for r in rows[start:]) - t0) / 1e3))
