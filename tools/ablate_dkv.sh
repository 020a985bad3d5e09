# timing-only ablations of the generated dK/dV stream (variant libraries built by hand: see profiles/r04/dev_log.md)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab_dkv; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in hip hip_ab_valu hip_ab_lds hip_ab_bardma hip_ab_all hip_ab_nomfma; do
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 12 > $OUT/$v.log 2>&1
f=$(ls $OUT/$v/*/*kernel_stats.csv | head -1)
echo "$v $(grep dkv64 $f | cut -d, -f4)"
done
