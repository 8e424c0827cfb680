#!/bin/bash
# Build libmi_gnina variants with different -D switches for vina.hip (A/B latency experiments).
# usage: tools/experiments/vina_variants.sh name1:"-DFOO=1 -DBAR=0" name2:"..."   -> gnina_amd/lib/variants/libmi_<name>.so
set -e
cd "$(dirname "$0")/../.."
python -c "from gnina_amd import build; build.build()"
mkdir -p gnina_amd/lib/variants
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $defs \
      -c gnina_amd/csrc/vina.hip -o gnina_amd/lib/variants/vina_$name.o
  objs=$(ls gnina_amd/lib/obj/*.o | grep -v "vina.hip.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gnina_amd/lib/variants/libmi_$name.so $objs gnina_amd/lib/variants/vina_$name.o
  echo built $name
done
