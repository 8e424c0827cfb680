#!/bin/bash
# Timing-experiment variants of the split-fp16 conv kernels into gnina_amd/lib/variants/ (run here, on the build box), then
# on the GPU box:  MI_GNINA_LIB=gnina_amd/lib/variants/libmi_gnina_h2x1.so python bench.py --no-configs --no-cpu-baseline
#   h2x1: no K loop (staging, barriers, epilogue only)   h2x2: staging without its global loads   h2x3: both
# The variants compute wrong results by construction; they only answer "where does the time go" (32-wide kernel only).
set -e
cd "$(dirname "$0")/.."
python -c "from gnina_amd import build; build.build()"
mkdir -p gnina_amd/lib/variants
for x in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -DMI_H2_EXPERIMENT=$x \
      -c gnina_amd/csrc/conv3d_h2.hip -o gnina_amd/lib/variants/conv3d_h2x$x.o
  objs=$(ls gnina_amd/lib/obj/*.o | grep -v "/conv3d_h2.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gnina_amd/lib/variants/libmi_gnina_h2x$x.so $objs gnina_amd/lib/variants/conv3d_h2x$x.o -ldl -lpthread
done
ls -la gnina_amd/lib/variants/
