#!/bin/bash
# Build timing-experiment variants of the conv kernel into gnina_amd/lib/variants/ (run here, on the build box),
# then on the GPU box:  MI_GNINA_LIB=gnina_amd/lib/variants/libmi_gnina_x1.so python bench.py --no-configs ...
#   x1: staging without its global loads   x2: everything but the K loop
#   x3: sparse K loop with free liveness bits, dead pairs skipped before their loads (ceiling of a bitmap-driven loop)
# The variants compute wrong results by construction; they only answer "where does the time go".
set -e
cd "$(dirname "$0")/.."
python -c "from gnina_amd import build; build.build()"
mkdir -p gnina_amd/lib/variants
for x in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -DMI_CONV_EXPERIMENT=$x \
      -c gnina_amd/csrc/conv3d.hip -o gnina_amd/lib/variants/conv3d_x$x.o
  objs=$(ls gnina_amd/lib/obj/*.o | grep -v "/conv3d.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gnina_amd/lib/variants/libmi_gnina_x$x.so $objs gnina_amd/lib/variants/conv3d_x$x.o
done
ls -la gnina_amd/lib/variants/
