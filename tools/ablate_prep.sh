#!/bin/bash
# Timing-only ablation of the K/V pre-pass (results meaningless): which part of its launch is what?
#   1 = no rho transform (unpack / pack / |k'|^2 kept); 2 = 1 + the V rows go out as they came in; 3 = 2 + no image stores;
#   4 = 2 + no loads (stale LDS); 5 = neither loads nor stores.
# build (here, on CPU):  bash tools/ablate_prep.sh build      -> gta_amd/csrc/libgta_hip_abl{1..5}.so (git-ignored)
# run (on the GPU box):  bash tools/ablate_prep.sh [workload]
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd gta_amd/csrc; make -j8 > /dev/null; mkdir -p build_var
  for n in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc -DGTA_PREP_ABL=$n -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -c gta_prep.hip -o build_var/gta_prep_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgta_hip_abl$n.so $(ls build/*.o | grep -v "gta_prep.hip.o\|gta_block\|gta_wgrad\|gta_gemm\|fwd64_diag") build_var/gta_prep_abl$n.o
  done
  ls -la libgta_hip_abl*.so
  exit 0
fi
WL=${1:-ms-enc}
for rep in 1 2; do
  python tools/time_prep.py $WL full 2>&1 | grep -v amdgpu.ids
  for n in 1 2 3 4 5; do GTA_HIP_LIB=$PWD/gta_amd/csrc/libgta_hip_abl$n.so python tools/time_prep.py $WL abl$n 2>&1 | grep -v amdgpu.ids; done
done
