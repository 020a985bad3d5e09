import sys; sys.path.insert(0,'.')
import torch
from tests import test_gpu_run_configs as T
for run in sorted(T.RUNS):
    for side in ("enc","dec"):
        dh, mixed, enc, dec = T.RUNS[run]
        scale = dh ** -0.5
        q,k,v,w,ex = T._inputs(dh, enc, dec, side, seed=sum(map(ord, run)) + (side == "dec"))
        dtype = torch.bfloat16 if mixed else torch.float32
        if mixed: q,k,v = (t.bfloat16().float() for t in (q,k,v))
        ref = T._oracle(q,k,v,w,ex,enc,dec,side,0.37,scale)
        got = T._hip(q,k,v,w,ex,enc,dec,side,0.37,scale,dtype,False)
        if got[4] is None: continue
        # sensitivity of the oracle's value to bf16-size relative perturbations of q, k, v
        g = torch.Generator().manual_seed(1)
        ds=[]
        for rep in range(3):
            qq,kk,vv = (t*(1+ (torch.rand(t.shape,generator=g)-0.5)*2**-8) for t in (q,k,v))
            r2 = T._oracle(qq,kk,vv,w,ex,enc,dec,side,0.37,scale)
            ds.append(abs(float(r2[4])-float(ref[4])))
        print(f"{run:24s} {side} got {float(got[4]):9.4f} ref {float(ref[4]):9.4f} err {abs(float(got[4])-float(ref[4])):.4f} input-noise sens {max(ds):.4f}")
