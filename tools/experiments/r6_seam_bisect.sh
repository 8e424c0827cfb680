#!/bin/bash
# why is the default ensemble's B = 1 call 600 us alone and 880-1,100 us inside the full bench?  which earlier phase does it?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for only in seam_b1 c5,seam_b1 c4,seam_b1 c3,seam_b1 c3_real,seam_b1 real_complex,seam_b1; do
  timeout 900 python bench.py --no-cpu-baseline --only $only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); a=d['also']['seam_b1']; print('$only', a['default_ensemble'])"
done
