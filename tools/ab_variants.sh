#!/bin/bash
# usage (on the GPU box): tools/ab_variants.sh NAME1 NAME2 ...   -- kernel cycles of each instrumented variant library
# (tools/build_variant.sh), interleaved over 3 passes; compare KERNEL CYCLES and the per-phase means, not microseconds.
cd "$(dirname "$0")/.."
[ -z "$GTA_TL_BOTH" ] && export GTA_TL_DEFAULT_ONLY=1
for r in 1 2 3; do
  for v in "$@"; do
    GTA_HIP_LIB=$PWD/gta_amd/csrc/libgta_var_$v.so python tools/bench_kernels.py timeline 2>&1 | python -c "
import sys,re
t=sys.stdin.read()
cyc=re.findall(r'KERNEL CYCLES ([\d.]+)k',t)
ph={k:re.findall(k+r'\s+mean\s+(\d+)',t) for k in ('start -> records staged, Q loads requested, barrier','rho_q','tile loop','epilogue','whole item')}
print('%-10s pass $r  kernel cycles %s   load %s rho %s loop %s epi %s item %s'%('$v','/'.join(cyc),ph['start -> records staged, Q loads requested, barrier'][0],ph['rho_q'][0],ph['tile loop'][0],ph['epilogue'][0],ph['whole item'][0]))
"
  done
done
