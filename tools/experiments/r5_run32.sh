#!/bin/bash
# the whole -m gpu suite with all 64 weight blobs on the box (the per-model parity tests against the reference's TorchScript outputs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
ls gnina_amd/weights | wc -l
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
