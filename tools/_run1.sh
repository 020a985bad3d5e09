python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/bench_kernels.py ab 2>&1 | grep -v amdgpu
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --train-steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['parity'], d['fwd_bwd']['ms_per_step'])"
