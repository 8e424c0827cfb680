#!/bin/bash
# round 5, GPU call 62: the whole -m gpu suite on the final library (weights of the seven committed models; the other 57
# blobs are skipped by their test when absent -- call 32 ran all 64)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
