"""How the granted clock and the step time develop from a cold start: the bench step (rep build + pre-pass + attention) back to back, reported per
chunk of steps (host clock per chunk; kernel time and granted clock of the chunk's last step from the dispatch events / stamps).
    python tools/clock_ramp.py [total_seconds] [workload]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gta_amd import native

total = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
wl = sys.argv[2] if len(sys.argv) > 2 else "ms-enc"
big = (sys.argv[3] if len(sys.argv) > 3 else "big") == "big"      # "small": keep 50-step chunks (a synchronisation every 10 ms) throughout
dev = torch.device("cuda", 0)
L = native.lib()
ps = bench.PlannedStep(wl, bench.WORKLOADS[wl][8], "bf16", dev, L, seed=1234, steps=1, kernel_samples=1, time_kernel=True)
torch.cuda.synchronize()
time.sleep(0.5)                                   # idle: the state a fresh process starts from
t_start = time.perf_counter()
chunk = 5
n = 0
while time.perf_counter() - t_start < total:
    t0 = time.perf_counter()
    for i in range(chunk - 1):
        ps.step()
    ps.step(0)                                    # (step index 0 is the sampled one of a 1-step plan: events + stamps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / chunk * 1e3
    kms, cyc, mhz = ps.kernel_times()
    n += chunk
    print(f"t = {1e3 * (time.perf_counter() - t_start):8.1f} ms  steps {n:6d}  ms/step {dt:.4f}  kernel {kms * 1e3:6.1f} us  {mhz:5.0f} MHz  {cyc / 1e3:6.1f}k cycles", flush=True)
    if n >= 40:
        chunk = 50
    if n >= 1000 and big:
        chunk = 1000
