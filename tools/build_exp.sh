#!/bin/bash
# Experiment library: the production objects + gta_abi.cpp built with -DGTA_ABLATE (the library then reads GTA_DBG from the environment
# at every call and hands it to the kernels as GtaFwdParams.dbg) -> gta_amd/csrc/libgta_hip_exp.so.  Use with GTA_HIP_LIB=... tools/exp_fwd2.py.
set -e
cd "$(dirname "$0")/../gta_amd/csrc"
make -j8 > /dev/null
mkdir -p build_var
/opt/rocm/bin/hipcc -DGTA_ABLATE -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -x hip -c gta_abi.cpp -o build_var/gta_abi_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgta_hip_exp.so $(ls build/*.o | grep -v "gta_abi.cpp.o\|gta_block\|gta_wgrad\|gta_gemm\|fwd64_diag") build_var/gta_abi_exp.o
ls -la libgta_hip_exp.so
