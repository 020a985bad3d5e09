import torch, time, sys
sys.path.insert(0,'.')
from gta_amd import native
from oracle import gta_oracle as O
g=torch.Generator().manual_seed(0)
E=O.random_extrinsics(32,5,g).cuda()
c=torch.rand(32,1280,2).cuda()
def t(fn,n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print("view reps L=2: %.1f us  L=0: %.1f us   so2 table: %.1f us"%(t(lambda: native.build_view_reps(E,2)), t(lambda: native.build_view_reps(E,0)), t(lambda: native.build_so2_table(c,6,1.0,1.0))))
