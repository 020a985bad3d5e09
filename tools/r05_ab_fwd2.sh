# A/B of the attention kernel alone (K'/V' images ready) at the dh = 64 workloads: libgta_hip_prev.so against libgta_hip.so, alternating
cd $GRAFT_REPO_ROOT
WLS=${WLS:-cl-dec,cl-enc,dit:rows32,dit}
for rep in 1 2; do
  for lib in libgta_hip_prev.so libgta_hip.so; do
    echo "---- $lib (pass $rep)"
    GTA_HIP_LIB=$PWD/gta_amd/csrc/$lib python tools/exp_fwd2.py $WLS 0 2 2>&1 | grep -v amdgpu.ids
  done
done
