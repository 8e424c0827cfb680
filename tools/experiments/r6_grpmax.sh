#!/bin/bash
# round 6: cap on the chunk groups of per-pose d16 launches (two Dense lanes share the chip: LDS per workgroup decides whether their workgroups co-reside)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for V in 0 2 3 4; do timeout 300 python tools/experiments/seam_b1_ensemble.py MI_GNINA_D16_GROUP_MAX=$V 2>&1 | head -1; done
for V in 0 3; do python tools/experiments/b1_grad_timeline.py dense MI_GNINA_D16_GROUP_MAX=$V; done
