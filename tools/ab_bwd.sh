# A/B of the backward kernels: libgta_hip_prev.so (the last commit) against libgta_hip.so, alternating on one box; kernel times under rocprofv3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab_bwd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
for v in prev new; do
LIB=$R/gta_amd/csrc/libgta_hip.so; [ $v = prev ] && LIB=$R/gta_amd/csrc/libgta_hip_prev.so
GTA_HIP_LIB=$LIB timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v$i -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --block-steps 0 --train-steps 20 ${AB_ARGS} > $OUT/$v$i.log 2>&1
f=$(ls $OUT/$v$i/*/*kernel_stats.csv | head -1)
python - $f $v$i $OUT/$v$i.log <<'PY'
import csv,sys,re
rows={r['Name']:float(r['AverageNs'])/1e3 for r in csv.DictReader(open(sys.argv[1]))}
g=lambda k: next((v for n,v in rows.items() if k in n),0)
fb=re.search(r'"fwd_bwd": \{"ms_per_step": ([0-9.]+)',open(sys.argv[3]).read())
print(sys.argv[2],'dkv %.1f dq %.1f prep %.1f attn %.1f kvprep %.1f | fwd_bwd %s'%(g('bwd_dkv'),g('bwd_dq'),g('bwd_prep'),g('attn64'),g('kv_prep'),fb.group(1) if fb else '?'))
PY
done
done
