#!/bin/bash
# round 6: gradient calls with their outputs through one pinned block and the sliced gmax backward -- tests, B = 1 latency, B = 256 throughput
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 1500 python -m pytest tests/test_gpu_gradient.py tests/test_gpu_cnn_refine.py tests/test_gpu_h2_range.py tests/test_host_adapter.py -m gpu -x -q 2>&1 | tail -4
python tools/experiments/b1_grad_timeline.py default2017
python tools/experiments/b1_grad_timeline.py default2017 MI_GNINA_OUT_COPY=1
python tools/experiments/b1_grad_timeline.py ensemble
python tools/experiments/b1_grad_timeline.py dense
python tools/experiments/grad_profile.py 2>&1 | grep "model\|gmax_backward\|sum of"
