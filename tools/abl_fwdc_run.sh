# developer tool: phase timings (tools/fwdc_phases.py) of the timing-only ablation builds of gta_fwd_cl.hip (tools/ab_fwdc.sh <bits>...)
cd $GRAFT_REPO_ROOT
for B in "$@"; do
  L=$PWD/gta_amd/csrc/libgta_hip.so
  [ "$B" != "0" ] && L=$PWD/gta_amd/csrc/libgta_var_fwdc$B.so
  echo "== ablation bits $B"
  GTA_HIP_LIB=$L python tools/fwdc_phases.py 2>&1 | grep " new "
done
