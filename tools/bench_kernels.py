#!/usr/bin/env python3
"""Per-kernel timing on MI355X (test/diagnostic tool): pre-pass vs flash-only vs fused, and the
fixed-vs-per-tile split (same Tq, varying Tk).  Usage: python tools/bench_kernels.py [workload] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gta_amd                      # noqa: E402
from gta_amd import native          # noqa: E402
from tests import _hip_cases as C   # noqa: E402


def setup(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype):
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=2)
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, f_dims)
    lay = lambda t: t.to(dtype).cuda().permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    return lay(q), lay(k), lay(v), packed, exd.get("gta_so3_degree", 0)


def time_call(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def raw_call(q, k, v, packed, f_dims, so3, flags, ws):
    B, H, Tq, dh = q.shape
    out = torch.empty(B, Tq, H, dh, device=q.device, dtype=q.dtype).permute(0, 2, 1, 3)
    lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    Nq = packed["vrep_q"].shape[1] if "vrep_q" in packed else 1
    Nk = packed["vrep_k"].shape[1] if "vrep_k" in packed else 1
    desc = native.make_desc(q, k, v, out, f_dims, so3, Nq, Nk, dh ** -0.5, flags)
    tc = torch.tensor([0.01], device=q.device) if f_dims.get("se3", 0) > 0 else None
    if ws is True:
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device=q.device, dtype=torch.uint8)
    fn = lambda: native.attn_fwd(desc, q, k, v, packed.get("vrep_q"), packed.get("vrep_k"), packed.get("cs_q"),
                                 packed.get("cs_k"), tc, None, out, lse, ws)
    return fn, out


def report(name, B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype=torch.bfloat16):
    q, k, v, packed, L = setup(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype)
    Tq, Tk, dh = Nq * Pq, Nk * Pk, sum(f_dims.values())
    flops = 4.0 * B * H * Tq * Tk * dh
    VT = native.FLAG_V_TRANSFORM
    fn_both, _ = raw_call(q, k, v, packed, f_dims, L, VT, True)
    fn_both()
    ws = torch.empty(1, device="cuda", dtype=torch.uint8)
    # flash only: reuse a filled workspace
    B_, H_, Tq_, dh_ = q.shape
    out = torch.empty(B, Tq, H, dh, device=q.device, dtype=q.dtype).permute(0, 2, 1, 3)
    desc = native.make_desc(q, k, v, out, f_dims, L, packed["vrep_q"].shape[1] if "vrep_q" in packed else 1,
                            packed["vrep_k"].shape[1] if "vrep_k" in packed else 1, dh ** -0.5, VT)
    ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
    fn_fill, _ = raw_call(q, k, v, packed, f_dims, L, VT, ws)
    fn_fill()
    fn_flash, _ = raw_call(q, k, v, packed, f_dims, L, VT | native.FLAG_KV_READY, ws)
    fn_flash8, _ = raw_call(q, k, v, packed, f_dims, L, VT | native.FLAG_KV_READY | native.FLAG_WG8, ws)
    t_flash8 = time_call(fn_flash8)
    fn_fused, _ = raw_call(q, k, v, packed, f_dims, L, VT | native.FLAG_FUSED_KV, None)
    t_both, t_flash, t_fused = time_call(fn_both), time_call(fn_flash), time_call(fn_fused)
    print(f"{name:10s} B={B:3d} Tq={Tq:5d} Tk={Tk:5d} dh={dh:3d} | two-stage {t_both*1e3:7.1f} us  flash-only {t_flash*1e3:7.1f} us "
          f"({flops/t_flash/1e9:6.1f} TF; 8-wave {t_flash8*1e3:7.1f} us)  prep ~{(t_both-t_flash)*1e3:6.1f} us | fused {t_fused*1e3:7.1f} us ({flops/t_fused/1e9:6.1f} TF)",
          flush=True)


MS = {"triv": 0, "se3": 48, "so3": 24, "so2": 24}
CL = {"se3": 32, "so2": 32}
if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    if which in ("all", "ms"):
        report("MS-enc", B, 8, 5, 256, 5, 256, MS, 6, 2)
    if which in ("all", "tiles"):
        for Pk in (64, 128, 256, 512):       # same Tq, varying number of key tiles
            report(f"MS Pk={Pk}", B, 8, 5, 256, 5, Pk, MS, 6, 2)
    if which == "ablate":
        q, k, v, packed, L = setup(B, 8, 5, 256, 5, 256, MS, 6, 2, torch.bfloat16)
        VT = native.FLAG_V_TRANSFORM
        out = torch.empty(B, 1280, 8, 96, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, MS, L, 5, 5, 96 ** -0.5, VT)
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
        raw_call(q, k, v, packed, MS, L, VT, ws)[0]()
        for nw8 in (0, 1):
            fn, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_KV_READY | (native.FLAG_WG8 if nw8 else 0), ws)
            for dbg, label in ((0, "baseline"), (8192, "stagger ~1.7k cycles"), (2048, "stagger long"), (4096, "setprio half"),
                               (8192 + 4096, "stagger+setprio"), (1, "no loop DMA"), (2, "no barrier"), (3, "no DMA, no barrier"),
                               (4, "no softmax math"), (8, "no PV"), (16, "no QK"), (12, "no softmax, no PV"),
                               (28, "no QK/softmax/PV (loop skeleton)"), (31, "empty loop"),
                               (31 + 32, "empty loop, no Q loads"), (31 + 64, "empty loop, no O stores"),
                               (31 + 128, "empty loop, no rep staging"), (31 + 256, "empty loop, no epilogue"),
                               (31 + 256 + 32 + 128, "empty loop, no epilogue/Q loads/reps"), (512, "bare launch")):
                os.environ["GTA_DBG"] = str(dbg)
                t = time_call(fn)
                print(f"  {'8-wave' if nw8 else '4-wave'} dbg={dbg:2d} {label:36s} {t*1e3:7.1f} us", flush=True)
            os.environ["GTA_DBG"] = "0"
    if which == "pipe":          # region cycle sums of the pipelined kernel (needs a -DGTA_ABLATE build)
        import ctypes
        q, k, v, packed, L = setup(B, 8, 5, 256, 5, 256, MS, 6, 2, torch.bfloat16)
        VT = native.FLAG_V_TRANSFORM
        out = torch.empty(B, 1280, 8, 96, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, MS, L, 5, 5, 96 ** -0.5, VT)
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
        raw_call(q, k, v, packed, MS, L, VT, ws)[0]()
        fn, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_KV_READY | native.FLAG_WG8, ws)
        nwg = B * 8 * 5
        for _ in range(3):
            fn()
        print(f"pipelined kernel: {time_call(fn)*1e3:.1f} us")
        prof = torch.zeros(nwg, 16, dtype=torch.int64, device="cuda")
        native.lib().gta_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
        torch.cuda.synchronize(); fn(); torch.cuda.synchronize()
        native.lib().gta_debug_set_profile_buffer(None)
        P = prof.cpu().double()
        print(f"  per WG cycles: prologue {(P[:,1]-P[:,0]).mean():8.0f}  tile0+loop {(P[:,2]-P[:,1]).mean():8.0f}  epilogue {(P[:,3]-P[:,2]).mean():8.0f}  total {(P[:,3]-P[:,0]).mean():8.0f}")
        names = ["wait+barrier", "dma issue", "R3 (QK || exp,pack,V reads)", "R1 (PV || reads, max)", "decide", "R2 (PV || exp)", "rescale"]
        for i2, nm in enumerate(names):
            print(f"   {nm:30s} {P[:,8+i2].mean():9.0f} cycles / WG   {P[:,8+i2].mean()/20:7.0f} per tile")
        print(f"   slow-path decisions per WG (of 19): mean {P[:,15].mean():.2f} max {P[:,15].max():.0f}")
    if which == "ctx":
        # the attention kernel in the bench's context: right behind the K/V pre-pass of the same step (instrumented build).
        # Prints its span and shader clock there and in a back-to-back loop of itself.
        import ctypes
        q, k, v, packed, L = setup(B, 8, 5, 256, 5, 256, MS, 6, 2, torch.bfloat16)
        VT = native.FLAG_V_TRANSFORM
        out = torch.empty(B, 1280, 8, 96, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, MS, L, 5, 5, 96 ** -0.5, VT)
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
        fn_both, _ = raw_call(q, k, v, packed, MS, L, VT, ws)
        fn_prep, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_PREP_ONLY, ws)
        fn_flash, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_KV_READY, ws)
        nwg = B * 8 * 10
        def measure(label, warm, pre):
            for _ in range(20):
                warm()
            prof = torch.zeros(nwg * 3, 8, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            pre()
            native.lib().gta_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
            fn_flash()
            torch.cuda.synchronize()
            native.lib().gta_debug_set_profile_buffer(None)
            P = prof.cpu().double()[:nwg]
            real = P[:, 6] - P[:, 5]
            ok = real > 0
            ghz = ((P[:, 4] - P[:, 0])[ok] / real[ok]).mean() * 0.1
            span = (P[:, 6].max() - P[:, 5].min()) / 100.0
            loop = (P[:, 3] - P[:, 2]).mean()
            print(f"{label:44s} span {span:7.1f} us  clock {ghz:.3f} GHz  kernel {span * ghz:7.1f}k cycles  "
                  f"tile loop {loop:8.0f} cycles / workgroup", flush=True)
        for _ in range(2):
            measure("attention kernel after itself", fn_flash, lambda: None)
            measure("attention kernel after the K/V pre-pass", fn_both, fn_prep)
            measure("after the pre-pass + 100 us idle", fn_both, lambda: (fn_prep(), torch.cuda._sleep(200000)))
    if which in ("timeline", "timeline_dt"):
        import ctypes
        if which == "timeline_dt":       # the dh = 64 pure-so2 shape (three workgroups per CU with the plain loop)
            H_, Nv, Pv, FD, so2_, so3_ = 16, 1, 1024, {"so2": 64}, 16, 0
        else:
            H_, Nv, Pv, FD, so2_, so3_ = 8, 5, 256, MS, 6, 2
        MS = FD
        q, k, v, packed, L = setup(B, H_, Nv, Pv, Nv, Pv, MS, so2_, so3_, torch.bfloat16)
        VT = native.FLAG_V_TRANSFORM
        dh_ = sum(MS.values())
        out = torch.empty(B, Nv * Pv, H_, dh_, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, MS, L, Nv, Nv, dh_ ** -0.5, VT)
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
        raw_call(q, k, v, packed, MS, L, VT, ws)[0]()
        fn, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_KV_READY, ws)
        for nw8 in (0,):
          nwg = B * H_ * ((Nv * Pv + 127) // 128)
          fn, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_KV_READY | (native.FLAG_WG8 if nw8 else 0), ws)
          for dbg in (0,):
            os.environ["GTA_DBG"] = str(dbg)
            for _ in range(3):
                fn()
            prof = torch.zeros(nwg * 3, 8, dtype=torch.int64, device="cuda")
            native.lib().gta_debug_set_profile_buffer(ctypes.c_void_p(prof.data_ptr()))
            torch.cuda.synchronize()
            fn()
            torch.cuda.synchronize()
            native.lib().gta_debug_set_profile_buffer(None)
            Pall = prof.cpu().double()
            P = Pall[:nwg]
            seg = Pall[nwg:].reshape(nwg, 2, 8)
            print(f"== {'8-wave' if nw8 else '4-wave'} attention kernel")
            if seg.abs().sum() > 0:
                for i, nm in enumerate(["QK^T MFMAs", "softmax + V reads", "PV MFMAs", "barrier", "waits (vmcnt/lgkm)", "K/Q read issue | DMA issue"]):
                    print(f"   {nm:26s} {seg[:,0,i].mean():9.0f} | {seg[:,1,i].mean():9.0f}   per tile {seg[:,0,i].mean()/20:7.0f} | {seg[:,1,i].mean()/20:7.0f}")
            t0 = P[:, 0].min()
            names = ["start->reps+Qloads landed", "Q rho+stage+frags", "main loop", "epilogue"]
            if not nw8:
                # s_memtime counts shader clocks, s_memrealtime the 100 MHz reference: their ratio is the clock the
                # kernel actually ran at
                real = (P[:, 6] - P[:, 5])
                ok = real > 0
                ghz = ((P[:, 4] - P[:, 0])[ok] / real[ok]).mean() * 0.1
                span_us = (P[:, 6].max() - P[:, 5].min()) / 100.0
                print(f"dbg={dbg}: kernel span {span_us:.1f} us by s_memrealtime; shader clock during the kernel {ghz:.3f} GHz; "
                      f"KERNEL CYCLES {span_us * ghz:.1f}k (span x clock: use this, not the span, to compare builds)")
            d = P[:, 7] - P[:, 0]
            print(f"   {'(start -> Q/cs loads issued)':28s} mean {d.mean():9.0f} ticks  min {d.min():9.0f} max {d.max():9.0f}")
            if not nw8:
                X = Pall[nwg:2 * nwg]
                for nm, a_, b_ in (("start -> before Q loads", P[:, 0], X[:, 0]), ("Q/cs loads issue", X[:, 0], X[:, 1]),
                                   ("record load issue", X[:, 1], X[:, 2]), ("tile DMA issue", X[:, 2], P[:, 7]),
                                   ("wait + record stores", P[:, 7], X[:, 3]), ("barrier", X[:, 3], P[:, 1])):
                    d = b_ - a_
                    print(f"     {nm:26s} mean {d.mean():9.0f}  min {d.min():9.0f} max {d.max():9.0f}")
            for i, nm in enumerate(names):
                d = P[:, i + 1] - P[:, i]
                print(f"   {nm:28s} mean {d.mean():9.0f} ticks  min {d.min():9.0f} max {d.max():9.0f}")
            st = (P[:, 0] - t0)
            order = torch.argsort(st)
            pass
            pass
        os.environ["GTA_DBG"] = "0"
    if which == "dt":            # dh = 64 shapes: these run three workgroups per CU (151 VGPRs, 51 KB LDS)
        report("DT", B, 16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0)
        report("CL-dec", B, 6, 3, 853, 2, 300, CL, 8, 0)
    if which in ("all", "others"):
        report("MS-dec", B, 8, 5, 512, 5, 256, MS, 6, 2)
        report("CL-enc", B, 6, 2, 300, 2, 300, CL, 8, 0)
        report("CL-dec", B, 6, 3, 853, 2, 300, CL, 8, 0)
        report("DT", B, 16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0)
