#!/bin/bash
# dense model at several LDS pad choices of the 16-wide split-fp16 kernel ("kb:y,x" ...): run on the GPU box
for cfg in ${@:-52:0,0 52:6,4 52:1,2 30:0,0 30:10,4}; do
  kb=${cfg%%:*}; pads=${cfg##*:}
  echo -n "LDS_KB=$kb pads=$pads  "; MI_GNINA_H2_LDS_KB=$kb MI_GNINA_H2_PADS=$pads python bench.py --model dense --no-configs --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']), 'n16 ms', round(sum(k['ms_per_step'] for k in d['kernels'] if 'to16' in k['kernel']),3), [k['ms_per_step'] for k in d['kernels'] if 's24' in k['kernel'] and 'to16' in k['kernel']])"
done
