"""GPU probe of libgta_block.so: every kernel and every GEMM epilogue against plain PyTorch, with timings.

    python tools/probe_block.py            # on an MI355X

Prints one line per check: what, max |err|, and (for GEMMs) microseconds against torch's own call.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gta_amd import native_block as nb  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def err(a, b):
    return (a.float() - b.float()).abs().max().item()


def try_(name, fn):
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        print(f"{name}: FAILED {type(e).__name__}: {str(e)[:200]}")


M, D, F = 40960, 768, 3072

# ---- LayerNorm
for xdt, ydt in ((torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)):
    x = (torch.randn(M, D, device=dev) * 2 + 0.3).to(xdt)
    g = torch.randn(D, device=dev) * 0.2 + 1
    b = torch.randn(D, device=dev) * 0.1
    y, mean, rstd = nb.ln_fwd(x, g, b, 1e-5, ydt)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), g, b, 1e-5)
    print(f"ln_fwd {xdt}->{ydt}: err {err(y, ref):.3e} (bf16 eps*max {ref.abs().max().item() * 2 ** -8:.3e}) "
          f"{timeit(lambda: nb.ln_fwd(x, g, b, 1e-5, ydt)):.1f} us, torch {timeit(lambda: torch.nn.functional.layer_norm(x, (D,), g.to(xdt), b.to(xdt), 1e-5)):.1f} us")
    for gdt in (ydt,):
        dy = torch.randn(M, D, device=dev).to(gdt)
        dres = torch.randn(M, D, device=dev).to(xdt)
        xr = x.float().requires_grad_(True)
        gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5).backward(dy.float())
        dx, dgam, dbet = nb.ln_bwd(dy, x, g, mean, rstd, dres)
        print(f"ln_bwd dy {gdt} x {xdt}: dx err {err(dx, xr.grad + dres.float()):.3e} dgamma {err(dgam, gr.grad):.3e} (max {gr.grad.abs().max().item():.1f}) "
              f"dbeta {err(dbet, br.grad):.3e}  {timeit(lambda: nb.ln_bwd(dy, x, g, mean, rstd, dres)):.1f} us")

# ---- GELU / colsum
for dt in (torch.float32, torch.bfloat16):
    x = (torch.randn(M, F, device=dev) * 1.5).to(dt)
    dy = torch.randn(M, F, device=dev).to(dt)
    y = nb.gelu_fwd(x)
    xr = x.float().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr)
    yr.backward(dy.float())
    dx = nb.gelu_bwd(dy, x)
    print(f"gelu {dt}: fwd err {err(y, yr):.3e} bwd err {err(dx, xr.grad):.3e}  fwd {timeit(lambda: nb.gelu_fwd(x)):.1f} us bwd {timeit(lambda: nb.gelu_bwd(dy, x)):.1f} us")
    cs = nb.colsum(x)
    print(f"colsum {dt} [{M},{F}]: err {err(cs, x.float().sum(0)):.3e} (max {x.float().sum(0).abs().max().item():.1f}) {timeit(lambda: nb.colsum(x)):.1f} us")
    x2 = x[:, :D]
    print(f"colsum strided {dt} [{M},{D}] ld {F}: err {err(nb.colsum(x2), x2.float().sum(0)):.3e}")

# ---- GEMMs
bf = torch.bfloat16
for cdt in (bf, torch.float32):
    tag = "bf16" if cdt == bf else "fp32"
    x = torch.randn(M, D, device=dev).to(cdt)
    W = (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(cdt)
    W1 = (torch.randn(F, D, device=dev) * D ** -0.5).to(cdt)
    W2 = (torch.randn(D, F, device=dev) * F ** -0.5).to(cdt)
    b1 = torch.randn(F, device=dev) * 0.1
    b2 = torch.randn(D, device=dev) * 0.1
    res = torch.randn(M, D, device=dev)

    def plain():
        y = nb.gemm(x, W, trans_b=True)
        ref = x.float() @ W.float().t()
        print(f"[{tag}] qkv GEMM: err {err(y, ref):.3e} (max {ref.abs().max().item():.2f})  {timeit(lambda: nb.gemm(x, W, trans_b=True)):.1f} us, "
              f"torch linear {timeit(lambda: torch.nn.functional.linear(x, W)):.1f} us")
    try_(f"[{tag}] plain", plain)

    def bias_res():
        h = torch.randn(M, F, device=dev).to(cdt)
        for bdt in (torch.float32, cdt):
            y = nb.gemm(h, W2, trans_b=True, epilogue=nb.EPI_BIAS, bias=b2.to(bdt), c=res, beta=1.0, out_dtype=torch.float32)
            ref = h.float() @ W2.float().t() + b2 + res
            print(f"[{tag}] ff2 GEMM + bias({bdt}) + residual -> fp32: err {err(y, ref):.3e}  "
                  f"{timeit(lambda: nb.gemm(h, W2, trans_b=True, epilogue=nb.EPI_BIAS, bias=b2.to(bdt), c=res, beta=1.0, out_dtype=torch.float32)):.1f} us, "
                  f"torch linear+add {timeit(lambda: torch.nn.functional.linear(h, W2, b2.to(cdt)) + res):.1f} us")
    try_(f"[{tag}] bias+residual", bias_res)

    def gelu_epi():
        pre_ref = x.float() @ W1.float().t() + b1
        y = nb.gemm(x, W1, trans_b=True, epilogue=nb.EPI_BIAS_GELU, bias=b1)
        e_erf = err(y, torch.nn.functional.gelu(pre_ref))
        e_tanh = err(y, torch.nn.functional.gelu(pre_ref, approximate="tanh"))
        print(f"[{tag}] ff1 GEMM + bias + GELU: err vs erf {e_erf:.3e}, vs tanh {e_tanh:.3e} (max {pre_ref.abs().max().item():.2f})  "
              f"{timeit(lambda: nb.gemm(x, W1, trans_b=True, epilogue=nb.EPI_BIAS_GELU, bias=b1)):.1f} us, "
              f"torch linear+gelu {timeit(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, W1, b1.to(cdt)))):.1f} us")
        aux = torch.empty(M, F, device=dev, dtype=cdt)
        y2 = nb.gemm(x, W1, trans_b=True, epilogue=nb.EPI_BIAS_GELU_AUX, bias=b1, aux=aux)
        print(f"[{tag}] ... with aux: y err vs no-aux {err(y2, y):.3e}, aux err vs pre {err(aux, pre_ref):.3e}  "
              f"{timeit(lambda: nb.gemm(x, W1, trans_b=True, epilogue=nb.EPI_BIAS_GELU_AUX, bias=b1, aux=aux)):.1f} us")
        return aux
    aux_holder = {}
    try_(f"[{tag}] gelu epilogue", lambda: aux_holder.setdefault("aux", gelu_epi()))

    def dgelu():
        aux = aux_holder.get("aux")
        if aux is None:
            aux = (x.float() @ W1.float().t() + b1).to(cdt)
        dout = torch.randn(M, D, device=dev).to(cdt)
        dh_ref = dout.float() @ W2.float()
        pr = aux.float().requires_grad_(True)
        torch.nn.functional.gelu(pr).backward(dh_ref)
        d_erf = pr.grad.clone()
        pr.grad = None
        torch.nn.functional.gelu(pr, approximate="tanh").backward(dh_ref)
        d_tanh = pr.grad
        for epi, nm in ((nb.EPI_DGELU, "DGELU"), (nb.EPI_DGELU_BGRAD, "DGELU_BGRAD")):
            def one():
                bg = torch.zeros(F, device=dev)
                y = nb.gemm(dout, W2, epilogue=epi, aux=aux, bias=bg if epi == nb.EPI_DGELU_BGRAD else None)
                msg = f"[{tag}] dgrad GEMM + {nm}: err vs erf {err(y, d_erf):.3e} vs tanh {err(y, d_tanh):.3e} (max {d_erf.abs().max().item():.2f})"
                if epi == nb.EPI_DGELU_BGRAD:
                    msg += f" bgrad err {err(bg, y.float().sum(0)):.3e} (max {y.float().sum(0).abs().max().item():.1f})"
                msg += f"  {timeit(lambda: nb.gemm(dout, W2, epilogue=epi, aux=aux, bias=bg if epi == nb.EPI_DGELU_BGRAD else None)):.1f} us"
                print(msg)
            try_(f"[{tag}] {nm}", one)
        print(f"[{tag}] torch dgrad + gelu backward: "
              f"{timeit(lambda: torch.ops.aten.gelu_backward(dout @ W2, aux)):.1f} us")
    try_(f"[{tag}] dgelu", dgelu)

    def wgrad():
        dout = torch.randn(M, D, device=dev).to(cdt)
        h = torch.randn(M, F, device=dev).to(cdt)
        ref = dout.float().t() @ h.float()
        for odt in (torch.float32, cdt):
            y = nb.gemm(dout, h, trans_a=True, out_dtype=odt)
            print(f"[{tag}] wgrad GEMM -> {odt}: err {err(y, ref):.3e} (max {ref.abs().max().item():.1f})  "
                  f"{timeit(lambda: nb.gemm(dout, h, trans_a=True, out_dtype=odt)):.1f} us, torch {timeit(lambda: dout.t() @ h):.1f} us")

            def bg():
                bgv = torch.zeros(D, device=dev, dtype=odt)
                y2 = nb.gemm(dout, h, trans_a=True, out_dtype=odt, epilogue=nb.EPI_BGRAD_A, bias=bgv)
                print(f"[{tag}] wgrad GEMM + BGRAD_A -> {odt}: err {err(y2, ref):.3e} bgrad err {err(bgv, dout.float().sum(0)):.3e} "
                      f"(max {dout.float().sum(0).abs().max().item():.1f})  {timeit(lambda: nb.gemm(dout, h, trans_a=True, out_dtype=odt, epilogue=nb.EPI_BGRAD_A, bias=bgv)):.1f} us")
            try_(f"[{tag}] BGRAD_A {odt}", bg)
            if odt == cdt:
                break
    try_(f"[{tag}] wgrad", wgrad)

    def dgrad():
        dq = torch.randn(M, 3 * D, device=dev).to(cdt)
        y = nb.gemm(dq, W)
        print(f"[{tag}] dgrad GEMM: err {err(y, dq.float() @ W.float()):.3e}  {timeit(lambda: nb.gemm(dq, W)):.1f} us, torch {timeit(lambda: dq @ W):.1f} us")
    try_(f"[{tag}] dgrad", dgrad)
    if cdt == torch.float32:
        break
