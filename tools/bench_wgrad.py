"""Time gta_wgrad (with the fused bias gradient) at the MSN layer shapes, 40 960 tokens: python tools/bench_wgrad.py (on an MI355X)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gta_amd import native_block as nb
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 40960
for n, k in ((2304, 768), (768, 768), (1536, 768), (768, 1536), (3072, 768)):
    G = torch.randn(M, n, device=dev).to(torch.bfloat16)
    X = torch.randn(M, k, device=dev).to(torch.bfloat16)
    t1 = timeit(lambda: nb.wgrad(G, X, True))
    fl = 2.0 * M * n * k
    print(f"dW[{n},{k}] over {M} tokens: gta_wgrad(+bias) {t1:7.1f} us = {fl/t1*1e-6:6.0f} TFLOP/s")
