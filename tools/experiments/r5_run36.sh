#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for kb in 52 100 150; do
  echo "== MI_GNINA_H2_LDS_KB=$kb"
  MI_GNINA_H2_LDS_KB=$kb timeout 600 python tools/experiments/seam_b1_breakdown.py dense_1_3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-1500
done
