# kernel times of several library variants (gta_amd/csrc/libgta_<name>.so), alternating on one box: tools/r04_abn.sh name1 name2 ...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/abn; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
for v in "$@"; do
GTA_HIP_LIB=$R/gta_amd/csrc/libgta_$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 16 > $OUT/$v$i.log 2>&1
f=$(ls $OUT/$v$i/*/*kernel_stats.csv | head -1)
echo "$v$i dkv $(grep dkv64 $f | cut -d, -f4 | cut -c1-6) dq $(grep dq64 $f | cut -d, -f4 | cut -c1-6)"
done
done
