# PMC counters of the backward kernels (separate passes, --kernel-trace only)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_bwd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-parity --block-steps 0 --steps 2 --warmup 1 --train-steps 3"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/sq1 -- python $R/bench.py $B > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $OUT/sq2 -- python $R/bench.py $B > $OUT/sq2.log 2>&1
python - $OUT <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
for d in ('sq1','sq2'):
    f=glob.glob(f'{out}/{d}/*/*counter_collection.csv')
    if not f: print(d,'no counter file', glob.glob(f'{out}/{d}/*/*')); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        for key in ('bwd_dkv','bwd_dq','attn64_items'):
            if key in k: acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in acc.items():
        print(d,k,{n:round(sum(v)/len(v)) for n,v in c.items()})
PY
