#!/bin/bash
# bench the default model at several LDS budgets of the split-fp16 kernels (MI_GNINA_H2_LDS_KB): run on the GPU box
for kb in ${@:-30 52}; do
  echo "LDS_KB=$kb"; MI_GNINA_H2_LDS_KB=$kb python bench.py --no-configs --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']), [(k['kernel'][:28], k['ms_per_step']) for k in d['kernels'] if 'conv' in k['kernel']])"
done
