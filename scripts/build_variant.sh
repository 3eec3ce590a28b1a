#!/bin/bash
# Development: build harry_amd/variants/libharry_amd_NAME.so with extra flags for the reconstruction kernels
# (scripts/build_variant.sh clocks -DHRY_CHAIN_CLOCKS; SRC=general scripts/build_variant.sh genclocks -DHRY_GEN_CLOCKS for
# general.hip); run it with HRY_LIB=harry_amd/variants/libharry_amd_NAME.so
set -e
cd "$(dirname "$0")/../harry_amd/csrc"
name=$1; shift
src=${SRC:-unpredict}
mkdir -p ../variants ../../build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function -Wno-unused-result \
    --offload-arch=gfx950 "$@" -c device/$src.hip -o ../../build/variants/${src}_$name.o
objs=$(ls ../../build/obj/host/*.o ../../build/obj/device/*.o ../../build/obj/api.o | grep -v /$src.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs ../../build/variants/${src}_$name.o -o ../variants/libharry_amd_$name.so
echo built harry_amd/variants/libharry_amd_$name.so
