#!/bin/bash
# which hardware queue do two scorers' kernels run on?  (two threads, B = 1, default ensemble; lanes off / on)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 200 python tools/experiments/seam_b1_ensemble.py MI_GNINA_NO_LANES=1
export TMPDIR=/tmp; cd /tmp
for o in MI_GNINA_NO_LANES=1 MI_GNINA_LANES=2; do
  rm -rf $R/gpurun_out/prof_q; mkdir -p $R/gpurun_out/prof_q
  Q_THREADS=2 rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/prof_q -o t -- python $R/tools/experiments/seam_b1_two.py $o > $R/gpurun_out/prof_q/log.txt 2>&1
  tail -2 $R/gpurun_out/prof_q/log.txt
  python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/prof_q/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = rows[len(rows)//2:]   # the threaded phase
q = collections.Counter((r["Queue_Id"], r.get("Stream_Id", "")) for r in rows)
print("$o queues (Queue_Id, Stream_Id) -> kernels:", dict(q))
PY
done
