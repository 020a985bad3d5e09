#!/bin/bash
# usage: tools/build_variant.sh NAME "extra hipcc flags"   -> gta_amd/csrc/libgta_var_NAME.so (instrumented build of gta_fwd2.hip
# with the extra flags, linked with the other objects of the -DGTA_ABLATE build).  Developer tool for A/B runs.
set -e
cd "$(dirname "$0")/../gta_amd/csrc"
NAME=$1; shift
mkdir -p build_var
/opt/rocm/bin/hipcc -DGTA_ABLATE -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $@ -c gta_fwd2.hip -o build_var/fwd2_$NAME.o
OBJS=$(ls build_ablate/*.o | grep -v gta_fwd2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgta_var_$NAME.so build_var/fwd2_$NAME.o $OBJS
echo built libgta_var_$NAME.so
