import sys, torch
sys.path.insert(0, '/root/repo')
from types import SimpleNamespace
import gta_amd
from tests import _golden as G, _hip_cases as C
from oracle import gta_oracle as O
for case in G.list_cases("op_"):
    d, meta = G.load("op_" + case)
    if not C.FUSED_OK(meta) or meta["f_dims"].get("se3", 0) == 0: continue
    for dtype in (torch.float32,):
        ex = G.extras_of(d, torch.float32, "cuda")
        q, k, v = (torch.from_numpy(d[n]).to(dtype).cuda().requires_grad_() for n in "qkv")
        tc = torch.tensor([float(d["trans_coeff"])], device="cuda", requires_grad=True)
        out, _ = gta_amd.multihead_geometric_transform_attention(q, k, v, attn_fn=SimpleNamespace(scale=float(d["scale"])), f_dims=meta["f_dims"], reps=ex, trans_coeff=tc, v_transform=meta["v_transform"])
        (out.float() * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
        # oracle decomposition of dtc into the three parts, in fp64
        exo = G.extras_of(d)
        ak = G.attn_kwargs_of(meta)
        reps = O.encoder_reps(ak, exo)
        if meta["cross"]: reps = O.decoder_reps(ak, exo, reps)
        print(f"{case:16s} dtc got {tc.grad.item():+.5f} ref {float(d['dtrans_coeff'][0]):+.5f}  cross={meta['cross']} vt={meta['v_transform']}")
