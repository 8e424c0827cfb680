#!/bin/bash
# round 5, GPU call 53: lanes with every voxel group voxelized before the first lane starts -- bits and B = 1 latency,
# at the default number of hardware queues and at 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for q in "" 8; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/experiments/lanes_diag3.py 300 2>&1 | tail -4
done
