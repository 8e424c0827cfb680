#!/bin/bash
# round 6: gradient calls on lanes -- tests, B = 1 latency with / without
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_gradient.py tests/test_gpu_cnn_refine.py tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -12
python tools/experiments/b1_grad_timeline.py ensemble
python tools/experiments/b1_grad_timeline.py ensemble MI_GNINA_NO_GRAD_LANES=1
python tools/experiments/b1_grad_timeline.py default2017
