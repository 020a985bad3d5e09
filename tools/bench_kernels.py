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
    fn_flash8, _ = raw_call(q, k, v, packed, f_dims, L, VT | native.FLAG_KV_READY | native.FLAG_PERSIST, ws)
    t_flash8 = time_call(fn_flash8)
    fn_fused, _ = raw_call(q, k, v, packed, f_dims, L, VT | native.FLAG_FUSED_KV, None)
    t_both, t_flash, t_fused = time_call(fn_both), time_call(fn_flash), time_call(fn_fused)
    print(f"{name:10s} B={B:3d} Tq={Tq:5d} Tk={Tk:5d} dh={dh:3d} | two-stage {t_both*1e3:7.1f} us  flash-only {t_flash*1e3:7.1f} us "
          f"({flops/t_flash/1e9:6.1f} TF; persistent grid {t_flash8*1e3:7.1f} us)  prep ~{(t_both-t_flash)*1e3:6.1f} us | fused {t_fused*1e3:7.1f} us ({flops/t_fused/1e9:6.1f} TF)",
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
    if which == "ab":
        # fair A/B of two flag sets of the attention kernel alone: alternate them, report the median of 7 rounds each
        shapes = {"MS-enc": (8, 5, 256, 5, 256, MS, 6, 2), "MS-dec": (8, 5, 512, 5, 256, MS, 6, 2), "CL-enc": (6, 2, 300, 2, 300, CL, 8, 0),
                  "CL-dec": (6, 3, 853, 2, 300, CL, 8, 0), "DT": (16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0),
                  "MS-so2": (8, 5, 256, 5, 256, {"triv": 72, "so2": 24}, 6, 0)}      # (no per-view arithmetic: what the se3 / so3 matvecs cost)
        VT = native.FLAG_V_TRANSFORM
        for name, (H_, Nq_, Pq_, Nk_, Pk_, FD, so2_, so3_) in shapes.items():
            q, k, v, packed, L = setup(B, H_, Nq_, Pq_, Nk_, Pk_, FD, so2_, so3_, torch.bfloat16)
            dh_ = sum(FD.values())
            out = torch.empty(B, Nq_ * Pq_, H_, dh_, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
            desc = native.make_desc(q, k, v, out, FD, L, Nq_, Nk_, dh_ ** -0.5, VT)
            ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
            raw_call(q, k, v, packed, FD, L, VT, ws)[0]()
            fns = {"default": raw_call(q, k, v, packed, FD, L, VT | native.FLAG_KV_READY, ws)[0],
                   "persistent": raw_call(q, k, v, packed, FD, L, VT | native.FLAG_KV_READY | native.FLAG_PERSIST, ws)[0]}
            res = {n: [] for n in fns}
            for _ in range(7):
                for n, fn in fns.items():
                    res[n].append(time_call(fn, n=10, warm=2))
            flops = 4.0 * B * H_ * Nq_ * Pq_ * Nk_ * Pk_ * dh_
            msg = "  ".join(f"{n}: median {sorted(r)[3]*1e3:7.1f} us min {min(r)*1e3:7.1f} ({flops/sorted(r)[3]/1e9:6.1f} TF)" for n, r in res.items())
            print(f"{name:8s} {msg}", flush=True)
    if which == "plans":
        # whole forward call, both execution plans, at the BASELINE shapes (what gta_attn_fwd's plan choice should follow)
        shapes = {"MS-enc": (8, 5, 256, 5, 256, MS, 6, 2), "MS-dec": (8, 5, 512, 5, 256, MS, 6, 2), "CL-enc": (6, 2, 300, 2, 300, CL, 8, 0),
                  "CL-dec": (6, 3, 853, 2, 300, CL, 8, 0), "DT": (16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0),
                  "CL-enc-1v": (6, 1, 300, 1, 300, CL, 8, 0), "MS-3v": (8, 3, 256, 3, 256, MS, 6, 2)}
        VT = native.FLAG_V_TRANSFORM
        for name, (H_, Nq_, Pq_, Nk_, Pk_, FD, so2_, so3_) in shapes.items():
            q, k, v, packed, L = setup(B, H_, Nq_, Pq_, Nk_, Pk_, FD, so2_, so3_, torch.bfloat16)
            fns = {"two-stage": raw_call(q, k, v, packed, FD, L, VT, True)[0],
                   "two-stage persistent": raw_call(q, k, v, packed, FD, L, VT | native.FLAG_PERSIST, True)[0],
                   "single-kernel": raw_call(q, k, v, packed, FD, L, VT | native.FLAG_FUSED_KV, None)[0]}
            res = {n: [] for n in fns}
            for _ in range(5):
                for n, fn in fns.items():
                    res[n].append(time_call(fn, n=10, warm=2))
            print(f"{name:10s} Tq={Nq_*Pq_:5d} Tk={Nk_*Pk_:5d} " + "  ".join(f"{n}: {sorted(r)[2]*1e3:7.1f} us" for n, r in res.items()), flush=True)
    if which == "prep":
        # the K/V pre-pass alone (FLAG_PREP_ONLY) at the BASELINE shapes: time and effective bandwidth on algorithmic bytes
        shapes = {"MS-enc": (8, 5, 256, 5, 256, MS, 6, 2), "CL-enc": (6, 2, 300, 2, 300, CL, 8, 0), "DT": (16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0),
                  "MS-so2": (8, 5, 256, 5, 256, {"triv": 72, "so2": 24}, 6, 0)}      # (no per-view arithmetic: what the se3 / so3 matvecs cost)
        VT = native.FLAG_V_TRANSFORM
        for name, (H_, Nq_, Pq_, Nk_, Pk_, FD, so2_, so3_) in shapes.items():
            q, k, v, packed, L = setup(B, H_, Nq_, Pq_, Nk_, Pk_, FD, so2_, so3_, torch.bfloat16)
            dh_ = sum(FD.values())
            out = torch.empty(B, Nq_ * Pq_, H_, dh_, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
            desc = native.make_desc(q, k, v, out, FD, L, Nq_, Nk_, dh_ ** -0.5, VT)
            ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
            fn = raw_call(q, k, v, packed, FD, L, VT | native.FLAG_PREP_ONLY, ws)[0]
            ts = sorted(time_call(fn, n=20, warm=3) for _ in range(5))
            byts = 2.0 * B * H_ * Nk_ * Pk_ * dh_ * 2 * 2          # read K,V + write K',V' (bf16)
            print(f"{name:8s} prep median {ts[2]*1e3:6.1f} us  min {ts[0]*1e3:6.1f} us   {byts/ts[2]/1e9:6.2f} TB/s on {byts/1e6:.0f} MB algorithmic", flush=True)
    if which == "ctx2":
        # the attention kernel timed by its own dispatch events (gta_debug_time_next_attention_kernel) in three contexts:
        # back to back with itself, behind the K/V pre-pass of the same call (the bench's order), behind pre-pass + rep build
        import ctypes
        Lb = native.lib()
        q, k, v, packed, L = setup(B, 8, 5, 256, 5, 256, MS, 6, 2, torch.bfloat16)
        VT = native.FLAG_V_TRANSFORM
        out = torch.empty(B, 1280, 8, 96, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, MS, L, 5, 5, 96 ** -0.5, VT)
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
        fn_both, _ = raw_call(q, k, v, packed, MS, L, VT, ws)
        fn_flash, _ = raw_call(q, k, v, packed, MS, L, VT | native.FLAG_KV_READY, ws)
        fn_both()
        n = 30
        evs = [(Lb.gta_debug_event_create(), Lb.gta_debug_event_create()) for _ in range(n)]
        def run(label, fn, pre=None):
            for _ in range(5):
                if pre: pre()
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a, b in evs:
                if pre: pre()
                Lb.gta_debug_time_next_attention_kernel(ctypes.c_void_p(a), ctypes.c_void_p(b))
                fn()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e6
            ks = sorted(Lb.gta_debug_event_elapsed_ms(ctypes.c_void_p(a), ctypes.c_void_p(b)) * 1e3 for a, b in evs)
            print(f"{label:44s} attention kernel median {ks[n // 2]:7.1f} us  min {ks[0]:7.1f}  max {ks[-1]:7.1f}   loop period {wall:7.1f} us", flush=True)
        import time
        for _ in range(2):
            run("attention after itself", fn_flash)
            run("attention behind the K/V pre-pass", fn_both)
            run("behind pre-pass, 300 us idle in front", fn_both, pre=lambda: torch.cuda._sleep(600000))
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
            native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
            fn_flash()
            torch.cuda.synchronize()
            native.lib().gta_debug_profile_next_attention_kernel(None, 0)
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
        # per work item s_memtime stamps of the attention kernel (instrumented -DGTA_ABLATE build): persistent grid vs one
        # workgroup per query tile, in one process.  Kernel cycles = span x clock: use that, not the span, to compare builds.
        import ctypes
        if which == "timeline_dt":       # the dh = 64 pure-so2 shape (three workgroups per CU)
            H_, Nv, Pv, FD, so2_, so3_ = 16, 1, 1024, {"so2": 64}, 16, 0
        else:
            H_, Nv, Pv, FD, so2_, so3_ = 8, 5, 256, MS, 6, 2
        q, k, v, packed, L = setup(B, H_, Nv, Pv, Nv, Pv, FD, so2_, so3_, torch.bfloat16)
        VT = native.FLAG_V_TRANSFORM
        dh_ = sum(FD.values())
        out = torch.empty(B, Nv * Pv, H_, dh_, device="cuda", dtype=torch.bfloat16).permute(0, 2, 1, 3)
        desc = native.make_desc(q, k, v, out, FD, L, Nv, Nv, dh_ ** -0.5, VT)
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device="cuda", dtype=torch.uint8)
        raw_call(q, k, v, packed, FD, L, VT, ws)[0]()
        nwg = B * H_ * ((Nv * Pv + 127) // 128)
        for rnd in range(2):
          for label, extra in ((("one workgroup per tile", 0),) if os.environ.get("GTA_TL_DEFAULT_ONLY") else
                               (("one workgroup per tile", 0), ("persistent grid", native.FLAG_PERSIST))):
            fn, _ = raw_call(q, k, v, packed, FD, L, VT | native.FLAG_KV_READY | extra, ws)
            for _ in range(3):
                fn()
            t_us = time_call(fn) * 1e3
            prof = torch.zeros(nwg, 8, dtype=torch.int64, device="cuda")
            native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
            torch.cuda.synchronize()
            fn()
            torch.cuda.synchronize()
            native.lib().gta_debug_profile_next_attention_kernel(None, 0)
            P = prof.cpu().double()
            if P.abs().sum() == 0:
                print(f"== {label}: {t_us:.1f} us (no stamps: not an instrumented build)")
                continue
            real = P[:, 6] - P[:, 5]
            ok = real > 0
            ghz = ((P[:, 4] - P[:, 0])[ok] / real[ok]).mean() * 0.1
            span_us = (P[:, 6].max() - P[:, 5].min()) / 100.0
            print(f"== {label} (round {rnd}): {t_us:.1f} us by events; span {span_us:.1f} us by s_memrealtime; shader clock {ghz:.3f} GHz; "
                  f"KERNEL CYCLES {span_us * ghz:.1f}k")
            for nm, a_, b_ in (("start -> records staged, Q loads requested, barrier", 0, 1), ("rho_q", 1, 2), ("tile loop", 2, 3),
                               ("epilogue", 3, 4), ("  (loop end -> epilogue loads + touches issued)", 3, 7), ("whole item", 0, 4)):
                d = P[:, b_] - P[:, a_]
                print(f"   {nm:52s} mean {d.mean():9.0f}  min {d.min():9.0f}  max {d.max():9.0f}")
            # by round: items ranked by start time (100 MHz stamps), 512 per round
            t0 = P[:, 5].min()
            order = torch.argsort(P[:, 5])
            nslot = 512
            for r0 in range(0, nwg, nslot):
                idx = order[r0:r0 + nslot]
                Q = P[idx]
                print(f"   round {r0 // nslot}: start {((Q[:, 5] - t0) / 100).mean():7.1f} us (spread {((Q[:, 5].max() - Q[:, 5].min()) / 100):6.1f})  "
                      f"end {((Q[:, 6] - t0) / 100).mean():7.1f} us (spread {((Q[:, 6].max() - Q[:, 6].min()) / 100):6.1f})  item cycles: "
                      f"prologue {(Q[:, 2] - Q[:, 0]).mean():7.0f}  loop {(Q[:, 3] - Q[:, 2]).mean():7.0f}  epilogue {(Q[:, 4] - Q[:, 3]).mean():7.0f}  "
                      f"total {(Q[:, 4] - Q[:, 0]).mean():7.0f}")
            # occupancy over time: how many items are in flight at 20 sample points
            ends = (P[:, 6] - t0) / 100
            starts = (P[:, 5] - t0) / 100
            T = ends.max()
            occ = [int(((starts <= x) & (ends > x)).sum()) for x in [T * i / 20 for i in range(1, 20)]]
            print(f"   items in flight at 5%..95% of the span: {occ}")
    if which == "dt":            # dh = 64 shapes: these run three workgroups per CU (151 VGPRs, 51 KB LDS)
        report("DT", B, 16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0)
        report("CL-dec", B, 6, 3, 853, 2, 300, CL, 8, 0)
    if which in ("all", "others"):
        report("MS-dec", B, 8, 5, 512, 5, 256, MS, 6, 2)
        report("CL-enc", B, 6, 2, 300, 2, 300, CL, 8, 0)
        report("CL-dec", B, 6, 3, 853, 2, 300, CL, 8, 0)
        report("DT", B, 16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0)
