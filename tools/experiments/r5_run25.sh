#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== concurrency test"
timeout 600 python -m pytest tests/test_gpu_concurrency.py -m gpu -x -q 2>&1 | tail -6
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
