#!/bin/bash
# final sanity on the final tree: smoke, the suite (lean push), the one-rank torchrun launch of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torchrun N=1:', d['value'], d['ms_per_step'], d.get('rccl'))"
