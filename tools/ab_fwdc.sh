#!/bin/bash
# developer tool: timing-only ablation builds of gta_fwd_cl.hip -> gta_amd/csrc/libgta_var_fwdc<bits>.so (linked with the production objects)
set -e
cd "$(dirname "$0")/../gta_amd/csrc"
mkdir -p build_var
for B in "$@"; do
  /opt/rocm/bin/hipcc -DGTA_FWDC_ABL=$B -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -c gta_fwd_cl.hip -o build_var/gta_fwd_cl_$B.o 2>&1 | grep -E "error" || true
  OBJS=$(ls build/*.o | grep -v "gta_fwd_cl\|gta_block\|gta_wgrad\|gta_gemm\|_diag")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgta_var_fwdc$B.so build_var/gta_fwd_cl_$B.o $OBJS
  echo built libgta_var_fwdc$B.so
done
