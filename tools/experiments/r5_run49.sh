#!/bin/bash
# round 5, GPU call 49: strict summation order for eval_intramolecular of a model with flexible residues (VERDICT r4 missing #7)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_vina_ref.py -m gpu -x -q -k "flexible or strict or final" 2>&1 | tail -15
