import sys, ctypes, torch
sys.path.insert(0, ".")
import gta_amd
from gta_amd import native, plan, synth
CL = {"se3": 32, "so2": 32}
def run(name, B,H,Nq,Pq,Nk,Pk, flags):
    qm,km,vm,ex,ak,cross = synth.attention_inputs(B,H,Nq,Pq,Nk,Pk,CL,8,0,seed=5)
    exd={kk:vv.cuda() for kk,vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak,exd)
    if cross: gta_amd.pre_compute_reps_decoder(ak,exd)
    pk=gta_amd.pack_reps(exd,CL)
    qd,kd,vd=(synth.as_projection(t, torch.bfloat16, "cuda") for t in (qm,km,vm))
    tc=torch.tensor([0.01],device="cuda")
    f=plan.ForwardPlan(qd,kd,vd,CL,so3_degree=0,Nq=Nq,Nk=Nk,flags=flags)
    args=(qd,kd,vd,pk.get("vrep_q"),pk.get("vrep_k"),pk.get("cs_q"),pk.get("cs_k"),tc)
    import time
    t0=time.time()
    while time.time()-t0<1.0:
        for _ in range(50): f(*args)
        torch.cuda.synchronize()
    n_it, rows_it = ctypes.c_int32(0), ctypes.c_int32(0)
    kname=(native.lib().gta_debug_attention_kernel(ctypes.byref(f.desc), ctypes.byref(n_it), ctypes.byref(rows_it)) or b"").decode()
    res=[]
    for rep in range(5):
        prof=torch.zeros(n_it.value,8,dtype=torch.int64,device="cuda")
        native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), n_it.value)
        f(*args); torch.cuda.synchronize()
        for _ in range(20): f(*args)
        P=prof.cpu().double()
        ok=P[:,4]>0
        P=P[ok]
        span=(P[:,4].max()-P[:,0].min()).item()
        item=(P[:,4]-P[:,0]).median().item()
        ph=[]
        if (P[:,2]>0).all() and (P[:,3]>0).all():
            ph=[(P[:,2]-P[:,0]).median().item(),(P[:,3]-P[:,2]).median().item(),(P[:,4]-P[:,3]).median().item()]
        mhz=(P[:,4].max()-P[:,0].min()).item()/((P[:,6].max()-P[:,5].min()).item()/100.0)
        res.append((span,item,ph,mhz))
    res.sort()
    span,item,ph,mhz=res[len(res)//2]
    print(f"{name:8s} {kname:18s} items {n_it.value:5d} kernel span {span/1e3:7.1f}k cycles, item median {item/1e3:6.2f}k, phases(prologue, loop, epilogue) {[round(x/1e3,2) for x in ph]}, {mhz:.0f} MHz, {span/mhz:.1f} us")
for flags,tag in ((0,"new"),(native.FLAG_ROWS32|native.FLAG_FWD2_GENERIC,"old")):
    run("cl-enc "+tag,32,6,2,300,2,300,flags)
    run("cl-dec "+tag,32,6,3,853,2,300,flags)
