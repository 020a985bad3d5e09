#!/usr/bin/env python3
"""developer tool: medians of the per-item stamps of an attention launch (slots: 0 start, 1, 2, 3, 7 as the kernel defines them, 4 end; shader
cycles) for bench workloads:   python tools/item_phases.py dit,cl-dec [persist]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gta_amd import native

wls = (sys.argv[1] if len(sys.argv) > 1 else "dit").split(",")
persist = len(sys.argv) > 2 and sys.argv[2] == "persist"
dev = torch.device("cuda", 0)
L = native.lib()
for name in wls:
    ps = bench.PlannedStep(name, bench.WORKLOADS[name][8], "bf16", dev, L, seed=1, steps=1, kernel_samples=1,
                           flags=native.FLAG_PERSIST if persist else 0, time_kernel=False)
    vq, vk, cq, ck = ps.build_reps()
    ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc)
    torch.cuda.synchronize()
    n_it, rows_it = ctypes.c_int32(0), ctypes.c_int32(0)
    kname = (L.gta_debug_attention_kernel(ctypes.byref(ps.fwd.desc), ctypes.byref(n_it), ctypes.byref(rows_it)) or b"").decode()

    def run():
        return ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc, flags_extra=native.FLAG_KV_READY)
    bench.precondition(run, 1.0)
    prof = torch.zeros(max(n_it.value, 1), 8, dtype=torch.int64, device=dev)
    L.gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
    run()
    torch.cuda.synchronize()
    P = prof.cpu().double()
    P = P[P[:, 4] > P[:, 0]]
    cyc, mhz = bench.kernel_clock(prof)
    out = [f"{name}: {kname}, {len(P)} items of {rows_it.value} rows, launch {cyc / 1e3:.1f}k cycles at {mhz:.0f} MHz; item median {float((P[:, 4] - P[:, 0]).median()) / 1e3:.2f}k"]
    for k in (1, 2, 3, 7):
        m = P[:, k] > P[:, 0]
        if k == 1 and "fwdc" in kname:
            continue
        if m.sum() > len(P) // 2:
            out.append(f"stamp{k} - start {float((P[m, k] - P[m, 0]).median()) / 1e3:.2f}k")
    m = P[:, 3] > P[:, 0]
    if m.any():
        out.append(f"end - stamp3 {float((P[m, 4] - P[m, 3]).median()) / 1e3:.2f}k")
    print("; ".join(out), flush=True)
