#!/bin/bash
# round 5, GPU call 55: lanes on by default -- the concurrency tests, the DLScorer adapter test, the ensemble tests of the
# parity suite, and the B = 1 seam numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 900 python -m pytest tests/test_gpu_concurrency.py tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ensemble or ragged or chunking" 2>&1 | tail -3
python - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from gnina_amd import capi, synth
capi.init(0)
print(json.dumps(bench.config_seam_b1(capi, synth), indent=1)[-900:])
PY
