#!/usr/bin/env python3
"""A/B of gta_fwd2_kernel experiment switches (GTA_DBG bits, read by the library at every call) at bench.py's workloads: the attention
kernel alone (K'/V' images ready), alternating over the settings on one box; event time, kernel cycles and granted clock from the per-item
stamps, output compared bit for bit with the first setting's.

    python tools/exp_fwd2.py cl-dec,cl-enc,dit 0,16,64,80 [reps] [variable = GTA_DBG]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gta_amd import native, plan, synth


def main():
    wls = (sys.argv[1] if len(sys.argv) > 1 else "cl-dec,cl-enc").split(",")
    settings = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,16").split(",")]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    var = sys.argv[4] if len(sys.argv) > 4 else "GTA_DBG"          # the environment variable the settings go to
    dev = torch.device("cuda", 0)
    L = native.lib()
    for wl in wls:
        rows32 = wl.endswith(":rows32")
        name = wl.split(":")[0]
        ps = bench.PlannedStep(name, bench.WORKLOADS[name][8], "bf16", dev, L, seed=1, steps=1, kernel_samples=1,
                               flags=native.FLAG_ROWS32 if rows32 else 0, time_kernel=False)
        vq, vk, cq, ck = ps.build_reps()
        os.environ[var] = "0"
        ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc)           # fills the workspace
        torch.cuda.synchronize()
        n_it, rows_it = ctypes.c_int32(0), ctypes.c_int32(0)
        kname = (L.gta_debug_attention_kernel(ctypes.byref(ps.fwd.desc), ctypes.byref(n_it), ctypes.byref(rows_it)) or b"").decode()
        prof = torch.zeros(max(n_it.value, 1), 8, dtype=torch.int64, device=dev)
        ref = None
        res = {s: [] for s in settings}
        fl = ps.flops()

        def run():
            return ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc, flags_extra=native.FLAG_KV_READY)
        for rep in range(reps):
            for s in settings:
                os.environ[var] = str(s)
                for _ in range(4):
                    run()
                torch.cuda.synchronize()
                bench.precondition(run, float(os.environ.get("PRE_S", "0.4")))       # the sustained clock (tools/clock_ramp.py)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 30 * 1e3
                prof.zero_()
                L.gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
                out = run()
                torch.cuda.synchronize()
                cyc, mhz = bench.kernel_clock(prof)
                P = prof.cpu().double()
                item = float((P[:, 4] - P[:, 0])[P[:, 6] > P[:, 5]].mean())
                if ref is None:
                    ref = out.clone()
                same = bool(torch.equal(out, ref))
                res[s].append((us, cyc, mhz, item, same))
        print(f"== {wl}: {kname}, {n_it.value} items of {rows_it.value} rows", flush=True)
        for s in settings:
            r = res[s]
            print(f"   {var}={s:6d}: " + "  ".join(f"{u:6.1f} us" for u, *_ in r) + f" | launch+gap time; stamps: {r[-1][1] / 1e3:7.1f}k cycles at {r[-1][2]:6.0f} MHz, "
                  f"item {r[-1][3] / 1e3:5.1f}k cycles; frac(events) {fl / (min(u for u, *_ in r) * 1e-6) / 2.5e15:.3f}; bit-identical: {all(x[4] for x in r)}", flush=True)
        del ps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
