#!/bin/bash
# round 6: throughput tile of the split-format block layers (MI_GNINA_D16_TILE, read when a model is loaded)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for T in "" 224 242 422 244 222 226 424; do
  if [ -z "$T" ]; then python tools/experiments/dense_throughput.py 2>&1 | tail -1; else MI_GNINA_D16_TILE=$T python tools/experiments/dense_throughput.py 2>&1 | tail -1 | sed "s/^/tile $T: /"; fi
done
