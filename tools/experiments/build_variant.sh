#!/bin/bash
# Build gnina_amd/lib/variants/libmi_gnina_<name>.so: the library with ONE source recompiled under extra flags
# (kernel experiments; select at run time with MI_GNINA_LIB=<path>).  usage: build_variant.sh <name> <source under csrc/> <flags...>
set -e
R=$(cd $(dirname $0)/../.. && pwd)
NAME=$1; SRC=$2; shift 2
mkdir -p $R/gnina_amd/lib/variants
OBJ=$R/gnina_amd/lib/variants/${NAME}_$(basename $SRC).o
# (the product's flags, gnina_amd/build.py FLAGS -- except -packed-fp32-ops, which a variant asks for itself: pass
#  -Xclang -target-feature -Xclang -packed-fp32-ops to build without packed-fp32 instructions like the product)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -x hip "$@" -c $R/gnina_amd/csrc/$SRC -o $OBJ
OBJS=$(ls $R/gnina_amd/lib/obj/*.o | grep -v "/$(basename $SRC).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gnina_amd/lib/variants/libmi_gnina_$NAME.so $OBJS $OBJ -ldl -lpthread
rm -f $OBJ
echo $R/gnina_amd/lib/variants/libmi_gnina_$NAME.so
