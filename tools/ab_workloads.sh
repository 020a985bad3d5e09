for w in dit ms-dec; do for m in prepass prepass_rows32; do
timeout 200 python bench.py --workload $w --kv-mode $m --no-cpu-baseline --train-steps 0 --block-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w $m', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['kernel_ms']*1e3,1), round(r['frac'],3), r.get('sclk_mhz'), d['parity']['parity_max_abs'])"
done; done
