#!/bin/bash
# Development build of the 64-rows-per-wave attention kernel: gen_attn64.py --dev variants (schedules, timing-only ablations)
# in gta_amd/csrc/libgta_hip_dev.so (use with GTA_HIP_LIB=... and GTA_ATTN64_VARIANT=n).  Everything but the loop statement is
# the production code (the per-item stamps are always compiled in), so the per-phase cycles it reports are the shipped kernel's.
set -e
cd "$(dirname "$0")/../gta_amd/csrc"
mkdir -p build_var
python3 gen_attn64.py --dev --out build_var/gta_attn64_dev.inc "$@"
make -s >/dev/null
/opt/rocm/bin/hipcc -DGTA_ATTN64_DEV $EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -c gta_fwd64.hip -o build_var/gta_fwd64_dev.o
OBJS=$(ls build/*.o | grep -v "gta_fwd64\|gta_block\|gta_wgrad\|gta_gemm")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgta_hip_dev.so build_var/gta_fwd64_dev.o $OBJS
echo built libgta_hip_dev.so
