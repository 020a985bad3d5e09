# kernel sequence of one fused block layer (forward + backward) under rocprofv3: every launch of the LAST step with its duration and the idle time in front of it
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_block; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python $R/tools/bench_block.py --mode fused --no-launch-count --iters 400 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
f=$(ls $OUT/t/*/*kernel_trace.csv | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last forward+backward step: from the last ln_fwd that is followed by a gta_bwd kernel back to its start
names=[r['Kernel_Name'] for r in rows]
last_bwd=max(i for i,n in enumerate(names) if 'gta_bwd_prep' in n)
start=max(i for i in range(last_bwd) if 'ln_fwd' in names[i] and not any('ln_fwd' in names[j] for j in range(i+1,last_bwd)) is False) if False else None
# simpler: walk back from last_bwd to the second ln_fwd before it (a layer has two LayerNorms in its forward)
cnt=0; i=last_bwd
while i>0:
    if 'ln_fwd' in names[i]:
        cnt+=1
        if cnt==2: break
    i-=1
start=i
end=len(rows)
prev_end=int(rows[start-1]['End_Timestamp'])
tot=0; gaps=0
for r in rows[start:end]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name']
    n=n.replace('void ','').replace('(anonymous namespace)::','')[:78]
    print('%-78s %8.1f us  gap %6.1f' % (n,(e-s)/1e3,(s-prev_end)/1e3))
    tot+=(e-s)/1e3; gaps+=max(0,(s-prev_end)/1e3); prev_end=e
print('launches',end-start,'kernels %.1f us gaps %.1f us'%(tot,gaps))
PY
