"""Randomised layouts, shapes, dtypes and modes through gta_rep_apply / gta_rep_apply_bwd; prints a digest per case.  Run once per library
(GTA_HIP_LIB) with a dump file and compare with tools/cmp_apply.py: the LDS-staged kernels (r04) against the direct ones of an older build run
the same arithmetic per row and agree to fp32 rounding (measured: <= 1.2e-7 of the tensor's max over 90 random cases -- hipcc contracts a few
multiply-adds differently in the two forms; each library is run-to-run bit-reproducible).

    GTA_HIP_LIB=.../libgta_hip_prev.so python tools/stress_apply.py 1 40 a.pt;  python tools/stress_apply.py 1 40 b.pt;  python tools/cmp_apply.py a.pt b.pt"""
import hashlib
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gta_amd import native  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dump = sys.argv[3] if len(sys.argv) > 3 else None          # optional: a .pt file of every case's outputs (numeric comparison across libraries)
saved = []
random.seed(seed)
torch.manual_seed(seed)
dev = "cuda"


def digest(t):
    return hashlib.sha1(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    euclid = random.random() < 0.25
    L = random.choice([0, 1, 2])
    f = {"triv": random.choice([0, 0, 3, 8]), "se3": (3 if euclid else 4) * random.choice([0, 2, 8, 12]),
         "so3": (8 if L == 2 else 3) * random.choice([0, 1, 3]) if L else 0, "so2": 2 * random.choice([0, 4, 12, 16]),
         "t2": 3 * random.choice([0, 0, 2])}
    dh = sum(f.values())
    if dh == 0:
        continue
    B, H, N = random.choice([1, 2, 3]), random.choice([1, 2, 5]), random.choice([1, 2, 3])
    P = random.choice([1, 7, 33, 100, 130])
    T = N * P
    dt = random.choice([torch.float32, torch.bfloat16])
    mode = random.choice([0, 1, 2])
    x = torch.randn(B, T, H, dh, device=dev).to(dt).permute(0, 2, 1, 3)          # [B,H,T,dh] views of token-major memory
    dy = torch.randn(B, T, H, dh, device=dev).to(dt).permute(0, 2, 1, 3)
    y = torch.empty(B, T, H, dh, device=dev, dtype=dt).permute(0, 2, 1, 3)
    dx = torch.empty(B, T, H, dh, device=dev, dtype=dt).permute(0, 2, 1, 3)
    vrep = torch.randn(B, N, native.VREP_STRIDE, device=dev)
    nb = f["so2"] // 2
    ang = torch.randn(B, T, max(nb, 1), device=dev)
    cs = torch.stack([ang.cos(), ang.sin()], -1).reshape(B, T, -1).contiguous()
    coord = torch.rand(B, T, 2, device=dev)
    tc = torch.tensor([0.37], device=dev)
    flags = native.FLAG_V_TRANSFORM | (native.FLAG_EUCLID if euclid else 0)
    desc = native.make_desc(x, x, x, y, f, L, N, N, dh ** -0.5, flags)
    kb = torch.empty(B, H, (T + 63) // 64 * 64, device=dev) if (euclid and mode == 1) else None
    native.rep_apply(desc, mode, x, vrep, cs, coord, tc, y, kb, 0.5)
    rows = torch.empty(B, H, T, device=dev)
    dkb = torch.randn(B, H, (T + 63) // 64 * 64, device=dev) if (euclid and mode == 1) else None
    native.rep_apply_bwd(desc, mode, x, dy, vrep, cs, coord, tc, dx, rows, dkey_bias=dkb, bias_scale=0.5)
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all())
    if dump:
        saved.append((it, y.float().cpu(), dx.float().cpu(), rows.cpu(), None if kb is None else kb[..., :T].cpu()))
    print(it, f, L, B, H, N, P, str(dt)[6:], mode, int(euclid), digest(y), digest(dx), digest(rows), "" if kb is None else digest(kb[..., :T]), "finite" if ok else "NONFINITE", flush=True)
if dump:
    torch.save(saved, dump)
