"""One MSN-Hard encoder layer (d = 768 = 8 x 96, mlp 1536, T = 1280 tokens, B scenes) as the fused block of
gta_amd.fused against the module-by-module path, bf16 autocast, forward and forward + backward.

    python tools/bench_block.py [--batch 32] [--mlp 3072] [--layers 1]

Prints microseconds per layer and the kernel launches per layer (torch profiler) for both paths.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gta_amd  # noqa: E402
from gta_amd import layers, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--mlp", type=int, default=1536, help="MSN gta_so3: mlp_dim = attdim * 2 (encoder.py) = 1536")
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=0.0, help="the reference configs train with 0.01 (forward+backward leg only)")
    ap.add_argument("--mode", default="both", choices=["both", "fused", "modules"])
    ap.add_argument("--no-launch-count", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    f_dims = {"triv": 0, "se3": 48, "so3": 24, "so2": 24}
    ak = {"f_dims": f_dims, "so2": 6, "so3": 2, "max_freq_h": 1, "max_freq_w": 1}
    tr = gta_amd.Transformer(768, args.layers, 8, 96, args.mlp, args.dropout, True, None, False, {"method": {"name": "gta", "args": ak}}).to(dev)
    B, V, hw = args.batch, 5, 16
    gen = torch.Generator().manual_seed(1)
    ex = {"input_transforms": synth.random_extrinsics(B, V, gen).to(dev), "input_coord": torch.rand(B, V, hw, hw, 2, generator=gen).to(dev)}
    gta_amd.pre_compute_reps_encoder(ak, ex)
    x0 = torch.randn(B, V * hw * hw, 768, device=dev)

    def fwd():
        tr.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return tr(x0, None, ex)

    def fwd_bwd():
        tr.train()
        x = x0.requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = tr(x, None, ex)
        y.backward(x0)
        for p in tr.parameters():
            p.grad = None
        x0.grad = None

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.iters * 1e3 / args.layers

    def launches(fn):
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        names = {}
        for e in evs:
            names[e.name] = names.get(e.name, 0) + 1
        return len(evs) / args.layers, names

    modes = [m for m in (("modules", False), ("fused", True)) if args.mode in ("both", m[0])]
    for label, on in modes * 2:
        tr.fused_blocks = on
        print(f"{label:8s} B={B} mlp={args.mlp}: forward {timeit(fwd):8.1f} us/layer   forward+backward {timeit(fwd_bwd):8.1f} us/layer")
    for label, on in ([] if args.no_launch_count else modes):
        tr.fused_blocks = on
        for what, fn in (("forward", fwd), ("forward+backward", fwd_bwd)):
            try:
                n, names = launches(fn)
                top = sorted(names.items(), key=lambda kv: -kv[1])
                print(f"{label:8s} {what}: {n:.0f} kernels per layer")
                for k, v in top:
                    print(f"      {v:3d} x {k[:110]}")
            except Exception as e:  # noqa: BLE001
                print(f"{label} {what}: profiler unavailable ({type(e).__name__}: {str(e)[:100]})")
    tr.fused_blocks = True


if __name__ == "__main__":
    main()
