#!/bin/bash
# usage: [FILE=gta_bwd.hip] tools/build_variant.sh NAME "extra hipcc flags"   -> gta_amd/csrc/libgta_var_NAME.so: an instrumented
# (-DGTA_ABLATE) build of FILE (default gta_fwd2.hip) with the extra flags, linked with the other objects of the
# -DGTA_ABLATE build.  Developer tool for A/B runs (tools/ab_variants.sh).
set -e
cd "$(dirname "$0")/../gta_amd/csrc"
NAME=$1; shift
FILE=${FILE:-gta_fwd2.hip}
BASEFLAGS="-fno-slp-vectorize"
[ "$FILE" != "gta_fwd2.hip" ] && BASEFLAGS=""
mkdir -p build_var
/opt/rocm/bin/hipcc -DGTA_ABLATE -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $BASEFLAGS $@ -c $FILE -o build_var/${FILE%.hip}_$NAME.o
OBJS=$(ls build_ablate/*.o | grep -v ${FILE%.hip} | grep -v "gta_block\|gta_wgrad\|gta_gemm")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgta_var_$NAME.so build_var/${FILE%.hip}_$NAME.o $OBJS
echo built libgta_var_$NAME.so
