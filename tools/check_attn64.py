#!/usr/bin/env python3
"""GPU check of the 64-rows-per-wave attention kernel (gta_fwd64.hip) against the 32-rows-per-wave one (GTA_FLAG_ROWS32)
and the oracle, with error localisation, plus an alternating A/B timing.  Developer tool.
usage: python tools/check_attn64.py [tiny|parity|time|all]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gta_amd                      # noqa: E402
from gta_amd import native          # noqa: E402
from tests import _hip_cases as C   # noqa: E402

MS = {"triv": 0, "se3": 48, "so3": 24, "so2": 24}
ROWS32 = 1 << 11


def build(B, H, Nq, Pq, Nk, Pk, dtype, seed=2, qmul=None, layout=(MS, 6, 2)):
    q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, layout[0], layout[1], layout[2], dtype, seed=seed)
    if qmul is not None:
        q = q.clone()
        q[B - 1] *= qmul              # hot logits in the last scene: the lazy softmax must rebase
        q[0, :, 40:90] *= qmul
    exd = {kk: vv.cuda() for kk, vv in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, exd)
    if cross:
        gta_amd.pre_compute_reps_decoder(ak, exd)
    packed = gta_amd.pack_reps(exd, layout[0])
    lay = lambda t: t.to(dtype).cuda().permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    return (q, k, v, ex, ak, cross), (lay(q), lay(k), lay(v), packed, exd.get("gta_so3_degree", 0), layout[0])


def run(dev, flags, ws=None, want_lse=False):
    q, k, v, packed, L, f_dims = dev
    B, H, Tq, dh = q.shape
    out = torch.zeros(B, Tq, H, dh, device=q.device, dtype=q.dtype).permute(0, 2, 1, 3)
    lse = torch.zeros(B, H, Tq, device=q.device, dtype=torch.float32)
    Nq, Nk = (packed["vrep_q"].shape[1], packed["vrep_k"].shape[1]) if "vrep_q" in packed else (1, 1)
    desc = native.make_desc(q, k, v, out, f_dims, L, Nq, Nk, dh ** -0.5, native.FLAG_V_TRANSFORM | flags)
    tc = torch.tensor([0.01], device=q.device) if f_dims.get("se3", 0) else None
    if ws is None:
        ws = torch.empty(native.attn_fwd_workspace_bytes(desc), device=q.device, dtype=torch.uint8)
    fn = lambda: native.attn_fwd(desc, q, k, v, packed.get("vrep_q"), packed.get("vrep_k"), packed.get("cs_q"), packed.get("cs_k"), tc,
                                 None, out, lse, ws)
    return fn, out, lse, ws


def localise(a, b, tag):
    d = (a.float() - b.float()).abs()          # [B,H,T,dh]
    B, H, T, dh = d.shape
    print(f"    {tag}: max {d.max().item():.3e}  (|ref| max {b.float().abs().max().item():.3e})")
    if d.max().item() == 0:
        return
    per_b = d.amax(dim=(1, 2, 3)).tolist()
    per_h = d.amax(dim=(0, 2, 3)).tolist()
    print("      per scene  ", " ".join(f"{x:.1e}" for x in per_b))
    print("      per head   ", " ".join(f"{x:.1e}" for x in per_h))
    rows = d.amax(dim=(0, 1, 3))               # [T]
    nb = (T + 31) // 32
    blk = [rows[32 * i:32 * i + 32].max().item() for i in range(nb)]
    print("      per 32-row block (first 16):", " ".join(f"{x:.1e}" for x in blk[:16]))
    it = [max(blk[8 * i:8 * i + 8]) for i in range((nb + 7) // 8)]
    print("      per 256-row item:", " ".join(f"{x:.1e}" for x in it))
    ch = d.amax(dim=(0, 1, 2))
    print("      per 8-channel chunk:", " ".join(f"{ch[8 * i:8 * i + 8].max().item():.1e}" for i in range(dh // 8)))
    print("      nan/inf in a:", int((~torch.isfinite(a.float())).sum().item()))


def parity_case(name, B, H, Nq, Pq, Nk, Pk, dtype, qmul=None, oracle=True):
    host, dev = build(B, H, Nq, Pq, Nk, Pk, dtype, qmul=qmul)
    os.environ.pop("GTA_ATTN64_VARIANT", None)
    os.environ.pop("GTA_ATTN64_COAL", None)
    fn_new, o_new, l_new, ws = run(dev, 0)
    fn_new()
    torch.cuda.synchronize()
    fn_old, o_old, l_old, _ = run(dev, ROWS32, ws=None)
    fn_old()
    torch.cuda.synchronize()
    os.environ["GTA_ATTN64_VARIANT"] = "1"
    fn_pl, o_pl, l_pl, _ = run(dev, 0)
    fn_pl()
    torch.cuda.synchronize()
    os.environ.pop("GTA_ATTN64_VARIANT", None)
    os.environ.pop("GTA_ATTN64_COAL", None)
    print(f"== {name}: B={B} H={H} Tq={Nq * Pq} Tk={Nk * Pk} {dtype}")
    ok = True
    e_no = (o_new.float() - o_old.float()).abs().max().item()
    e_po = (o_pl.float() - o_old.float()).abs().max().item()
    e_l = (l_new - l_old).abs().max().item()
    refmax = o_old.float().abs().max().item()
    print(f"    new vs rows32: {e_no:.3e}   plain vs rows32: {e_po:.3e}   lse new vs rows32: {e_l:.3e}   |out| max {refmax:.3e}")
    tol = 3e-2 * refmax
    if not (e_no <= tol) or not torch.isfinite(o_new.float()).all():
        ok = False
        localise(o_new, o_old, "new - rows32")
    if not (e_po <= tol) or not torch.isfinite(o_pl.float()).all():
        ok = False
        localise(o_pl, o_old, "plain - rows32")
    if oracle:
        q, k, v, ex, ak, cross = host
        ref = C.oracle_forward(q, k, v, ex, ak, cross, 0.01)
        for nm, o in (("new", o_new), ("rows32", o_old)):
            st = C.err_stats(o.float().cpu(), ref)
            print(f"    {nm} vs oracle: max_abs {st['max_abs']:.3e} rel_rms {st['rel_rms']:.3e} ref_max {st['ref_max']:.3e}")
            if st["max_abs"] > 2.5e-2 * st["ref_max"] or st["rel_rms"] > 1.2e-2:
                ok = False
                if nm == "new":
                    localise(o_new.cpu(), ref, "new - oracle")
    print("    ->", "OK" if ok else "MISMATCH")
    return ok


def time_ab(name, B, H, Nq, Pq, Nk, Pk):
    _, dev = build(B, H, Nq, Pq, Nk, Pk, torch.bfloat16)
    fn_fill, _, _, ws = run(dev, 0)
    fn_fill()
    fns = {"attn64": run(dev, native.FLAG_KV_READY, ws=ws)[0], "rows32": run(dev, native.FLAG_KV_READY | ROWS32, ws=ws)[0]}
    os.environ["GTA_ATTN64_VARIANT"] = "1"
    plain = run(dev, native.FLAG_KV_READY, ws=ws)[0]
    res = {n: [] for n in list(fns) + ["plain"]}

    def t(fn, n=10, warm=2):
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
    for _ in range(7):
        os.environ.pop("GTA_ATTN64_VARIANT", None)
        for n, fn in fns.items():
            res[n].append(t(fn))
        os.environ["GTA_ATTN64_VARIANT"] = "1"
        res["plain"].append(t(plain, n=4, warm=1))
    os.environ.pop("GTA_ATTN64_VARIANT", None)
    os.environ.pop("GTA_ATTN64_COAL", None)
    flops = 4.0 * B * H * Nq * Pq * Nk * Pk * 96
    print(f"== time {name}: " + "   ".join(f"{n}: median {sorted(r)[3] * 1e3:7.1f} us min {min(r) * 1e3:7.1f} ({flops / sorted(r)[3] / 1e9:6.1f} TF)"
                                            for n, r in res.items()), flush=True)


def timeline(B=32):
    """per-item s_memtime stamps (instrumented build: GTA_HIP_LIB=.../libgta_hip_ablate.so)"""
    import ctypes
    _, dev = build(B, 8, 5, 256, 5, 256, torch.bfloat16)
    fn_fill, _, _, ws = run(dev, 0)
    fn_fill()
    for label, flags, rows in (("attn64", 0, 256), ("rows32", ROWS32, 128)):
        fn = run(dev, native.FLAG_KV_READY | flags, ws=ws)[0]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        nwg = B * 8 * (1280 // rows)
        prof = torch.zeros(nwg, 8, dtype=torch.int64, device="cuda")
        native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        native.lib().gta_debug_profile_next_attention_kernel(None, 0)
        P = prof.cpu().double()
        if P.abs().sum() == 0:
            print(f"== {label}: no stamps (not an instrumented build)")
            continue
        real = P[:, 6] - P[:, 5]
        ok = real > 0
        ghz = ((P[:, 4] - P[:, 0])[ok] / real[ok]).mean() * 0.1
        span_us = (P[:, 6].max() - P[:, 5].min()) / 100.0
        print(f"== {label}: span {span_us:.1f} us; shader clock {ghz:.3f} GHz; KERNEL CYCLES {span_us * ghz:.1f}k; items {nwg}")
        for nm, a_, b_ in (("start -> records staged, loads landed, barrier", 0, 1), ("rho_q", 1, 2), ("tile loop", 2, 3), ("epilogue", 3, 4), ("whole item", 0, 4)):
            d = P[:, b_] - P[:, a_]
            print(f"   {nm:48s} mean {d.mean():9.0f}  min {d.min():9.0f}  max {d.max():9.0f}")
        t0 = P[:, 5].min()
        order = torch.argsort(P[:, 5])
        nslot = 256 if rows == 256 else 512
        for r0 in range(0, nwg, nslot):
            Q = P[order[r0:r0 + nslot]]
            print(f"   round {r0 // nslot}: start {((Q[:, 5] - t0) / 100).mean():7.1f} us  end {((Q[:, 6] - t0) / 100).mean():7.1f} us  "
                  f"prologue {(Q[:, 2] - Q[:, 0]).mean():7.0f}  loop {(Q[:, 3] - Q[:, 2]).mean():7.0f}  epilogue {(Q[:, 4] - Q[:, 3]).mean():7.0f}  total {(Q[:, 4] - Q[:, 0]).mean():7.0f}")


def variants(which, B=32):
    """development library (tools/build_attn64_dev.sh, GTA_HIP_LIB=.../libgta_hip_dev.so): event time and per-item cycle stamps per
    variant at 20 and 40 key tiles: the slope is the steady-state step, the intercept everything else of an item"""
    import ctypes
    devs = {}
    for nk in (5, 10):
        _, dev = build(B, 8, 5, 256, nk, 256, torch.bfloat16)
        fn_fill, _, _, ws = run(dev, 0)
        fn_fill()
        torch.cuda.synchronize()
        devs[nk] = (run(dev, native.FLAG_KV_READY, ws=ws)[0], run(dev, native.FLAG_KV_READY | ROWS32, ws=ws)[0])
    nwg = B * 8 * 5

    def t(f, n=10, warm=2):
        for _ in range(warm):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def stamps(f, rows):
        n = B * 8 * (1280 // rows)
        prof = torch.zeros(n, 8, dtype=torch.int64, device="cuda")
        native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
        torch.cuda.synchronize()
        f()
        torch.cuda.synchronize()
        native.lib().gta_debug_profile_next_attention_kernel(None, 0)
        P = prof.cpu().double()
        real = P[:, 6] - P[:, 5]
        ok = real > 0
        ghz = ((P[:, 4] - P[:, 0])[ok] / real[ok]).mean().item() * 0.1 if ok.any() else 0.0
        span_us = ((P[:, 6].max() - P[:, 5].min()) / 100.0).item()
        ph = [(P[:, b_] - P[:, a_]).mean().item() for a_, b_ in ((0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (3, 7))]
        return ghz, span_us, ph
    def setvar(v):                  # "3" = generated variant 3; "0:0" = variant 0 without the coalesced item I/O (dev builds)
        a, b = (str(v).split(":") + [""])[:2]      # variant : coalesced item I/O (0 / 1; development builds)
        os.environ["GTA_ATTN64_VARIANT"] = a
        if b:
            os.environ["GTA_ATTN64_COAL"] = b
        else:
            os.environ.pop("GTA_ATTN64_COAL", None)
    times = {v: [] for v in which}
    times["rows32"] = []
    for _ in range(5):
        for v in which:
            setvar(v)
            times[v].append(t(devs[5][0]))
        times["rows32"].append(t(devs[5][1]))
    for v in which:
        setvar(v)
        ghz, span_us, ph = stamps(devs[5][0], 256)
        _, _, ph2 = stamps(devs[10][0], 256)
        r = sorted(times[v])
        slope = (ph2[2] - ph[2]) / 20
        print(f"variant {v}: {r[2]:7.1f} us median ({r[0]:.1f} min) | {span_us:6.1f} us x {ghz:.3f} GHz = {span_us * ghz:6.1f}k cyc | item: "
              f"load {ph[0]:5.0f} rho_q {ph[1]:5.0f} loop {ph[2]:6.0f} epi {ph[3]:5.0f} (request issued at +{ph[5]:5.0f}) total {ph[4]:6.0f} | STEP {slope:5.0f} cyc/tile, head+tail {ph[2] - 20 * slope:5.0f}", flush=True)
    os.environ.pop("GTA_ATTN64_VARIANT", None)
    os.environ.pop("GTA_ATTN64_COAL", None)
    ghz, span_us, ph = stamps(devs[5][1], 128)
    _, _, ph2 = stamps(devs[10][1], 128)
    r = sorted(times["rows32"])
    print(f"rows32   : {r[2]:7.1f} us median ({r[0]:.1f} min) | {span_us:6.1f} us x {ghz:.3f} GHz = {span_us * ghz:6.1f}k cyc | item(128 rows): "
          f"load {ph[0]:5.0f} rho_q {ph[1]:5.0f} loop {ph[2]:6.0f} epi {ph[3]:5.0f} total {ph[4]:6.0f} | STEP {(ph2[2] - ph[2]) / 20:5.0f} cyc/tile")


def phases(workload):
    """per-item phase cycles of the attention kernel at one of bench.py's workloads, 64-row kernel and 32-row kernel"""
    import ctypes
    import bench
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, B = bench.WORKLOADS[workload]
    _, dev = build(B, H, Nq, Pq, Nk, Pk, torch.bfloat16, layout=(f_dims, so2, so3))
    fn_fill, _, _, ws = run(dev, 0)
    fn_fill()
    torch.cuda.synchronize()
    for label, flags in (("attn64", 0), ("rows32", ROWS32)):
        fn = run(dev, native.FLAG_KV_READY | flags, ws=ws)[0]
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        prof = torch.zeros(B * H * ((Nq * Pq + 127) // 128), 8, dtype=torch.int64, device="cuda")
        native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        native.lib().gta_debug_profile_next_attention_kernel(None, 0)
        P = prof.cpu().double()
        P = P[P[:, 4] > 0]
        real = P[:, 6] - P[:, 5]
        ghz = ((P[:, 4] - P[:, 0]) / real).mean().item() * 0.1
        span = ((P[:, 6].max() - P[:, 5].min()) / 100.0).item()
        ph = [(P[:, b_] - P[:, a_]).mean().item() for a_, b_ in ((0, 1), (1, 2), (2, 3), (3, 4), (0, 4))]
        print(f"{workload} {label}: {us:7.1f} us | {len(P)} items | span {span:6.1f} us x {ghz:.3f} GHz = {span * ghz:6.1f}k cyc | item: load {ph[0]:6.0f} "
              f"rho_q {ph[1]:6.0f} loop {ph[2]:6.0f} epi {ph[3]:6.0f} total {ph[4]:6.0f}", flush=True)


def ramp(workload):
    """where the launch's time outside its items goes: workgroup begin -> first item start, item durations by position in the
    workgroup's walk, the spread of the workgroups' ends, and what the dispatch's own events add around all of it"""
    import ctypes
    import bench
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, B = bench.WORKLOADS[workload]
    _, dev = build(B, H, Nq, Pq, Nk, Pk, torch.bfloat16, layout=(f_dims, so2, so3))
    fn_fill, _, _, ws = run(dev, 0)
    fn_fill()
    torch.cuda.synchronize()
    fn = run(dev, native.FLAG_KV_READY, ws=ws)[0]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    n_items = B * H * ((Nq * Pq + 255) // 256)
    for rep in range(2):
        prof = torch.zeros(n_items, 8, dtype=torch.int64, device="cuda")
        native.lib().gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        us1 = e0.elapsed_time(e1) * 1000
        P = prof.cpu().double()
        G = int((P[:, 2] > 0).sum().item())                 # workgroups (rows of first items carry the begin stamp)
        k = torch.arange(n_items) // G
        rt0 = P[:G, 2].min()
        begin = (P[:G, 2] - rt0) / 100.0                     # us
        first = (P[:G, 5] - rt0) / 100.0
        end_wg = torch.stack([P[(torch.arange(n_items) % G) == g, 6].max() for g in range(G)])
        end = (end_wg - rt0) / 100.0
        ghz = ((P[:, 4] - P[:, 0]) / (P[:, 6] - P[:, 5])).mean().item() * 0.1
        cyc_first = (P[:G, 0] - P[:G, 1])
        print(f"{workload} run {rep}: back-to-back {us:6.1f} us per launch, this (profiled, alone) {us1:6.1f} us | {G} workgroups, {n_items} items, {ghz:.3f} GHz")
        print(f"  workgroup begin     : {begin.min():6.2f} .. {begin.max():6.2f} us (median {begin.median():.2f})")
        print(f"  first item starts   : {first.min():6.2f} .. {first.max():6.2f} us (median {first.median():.2f}); begin -> first item {cyc_first.median():.0f} cycles median, {cyc_first.max():.0f} max")
        print(f"  workgroup ends      : {end.min():6.2f} .. {end.max():6.2f} us (median {end.median():.2f})")
        for kk in range(int(k.max().item()) + 1):
            d = (P[k == kk, 4] - P[k == kk, 0])
            print(f"  item {kk} of the walk  : {d.median():7.0f} cycles median ({d.min():.0f} .. {d.max():.0f})")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    ok = True
    if which in ("tiny", "all"):
        ok &= parity_case("tiny", 1, 1, 1, 256, 1, 256, torch.bfloat16)
        ok &= parity_case("tiny-2items", 1, 2, 2, 256, 2, 256, torch.bfloat16)
    if which in ("parity", "all"):
        ok &= parity_case("ms-enc", 2, 8, 5, 256, 5, 256, torch.bfloat16)
        ok &= parity_case("ms-enc-tail", 2, 8, 5, 250, 5, 250, torch.bfloat16)
        ok &= parity_case("ms-dec", 1, 8, 5, 512, 5, 256, torch.bfloat16)
        ok &= parity_case("ms-enc-f32", 1, 8, 5, 256, 5, 256, torch.float32)
        ok &= parity_case("ms-enc-hot", 2, 8, 5, 256, 5, 256, torch.bfloat16, qmul=12.0)
        ok &= parity_case("ms-enc-hot-tail", 2, 4, 5, 250, 5, 250, torch.bfloat16, qmul=12.0)
        ok &= parity_case("ms-b32 (many items per workgroup)", 32, 8, 5, 256, 5, 256, torch.bfloat16, oracle=False)
    if which in ("time", "all"):
        time_ab("MS-enc B32", 32, 8, 5, 256, 5, 256)
        time_ab("MS-dec B32", 32, 8, 5, 512, 5, 256)
    if which == "timeline":
        timeline()
    if which == "phases":
        for w in (sys.argv[2] if len(sys.argv) > 2 else "dit,ms-dec").split(","):
            phases(w)
    if which == "ramp":
        ramp(sys.argv[2] if len(sys.argv) > 2 else "ms-enc")
    if which == "variants":
        variants((sys.argv[2] if len(sys.argv) > 2 else "0,2,3,4,5,6,7,8").split(","))
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
