#!/usr/bin/env python3
"""The energy floor of the headline attention launch: tests/probes/probe_mfma_power.hip -- this launch's 4.9 M
v_mfma_f32_32x32x16_bf16 on random bf16 fragments, nothing else -- timed ALONE and INSIDE the bench step (in the attention kernel's place,
behind the real rep build + K/V pre-pass of every step), with the matrix pipe's duty cycle lowered by s_nop gaps, and with all-zero operands.

    python tools/probe_mfma_power.py [--reps 30] > gpurun_out/probe_mfma_power.json

Per variant: microseconds per launch (HIP events over back-to-back launches when alone; the 100-MHz stamps' span when inside the step),
shader cycles per launch, granted clock, and what `roofline.frac` would read for 161.06 GFLOP in that time."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOPS = 4.0 * 32 * 8 * 1280 * 1280 * 96
PEAK = 2500e12


def stamps_stats(st):
    S = st.cpu().double()
    cyc = S[:, 1] - S[:, 0]
    real = S[:, 3] - S[:, 2]
    mhz = float((cyc / real).mean()) * 100.0
    span_us = float(S[:, 3].max() - S[:, 2].min()) / 100.0
    return {"span_us": span_us, "sclk_mhz": mhz, "wg_cycles_mean": float(cyc.mean()), "wg_cycles_max": float(cyc.max()),
            "launch_cycles": span_us * mhz}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--precondition-s", dest="pre", type=float, default=0.0,
                    help="seconds of the variant's own launches (alone) / of the step (in the step) in front of each timed figure: the sustained "
                         "clock instead of the first milliseconds after idle (tools/clock_ramp.py)")
    args = ap.parse_args()
    so = os.path.join(ROOT, "tests", "probes", "libprobe_mfma_power.so")
    P = ctypes.CDLL(so)
    P.probe_mfma_power_launch.restype = ctypes.c_int
    P.probe_mfma_power_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = torch.randint(0, 2 ** 31 - 1, (P.probe_mfma_power_rnd_words(),), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    rnd ^= (torch.randint(0, 2, rnd.shape, generator=g, dtype=torch.int32) << 31).to(dev)     # (the sign bits too)
    sink = torch.zeros(256, device=dev)
    GRID, ITEMS = 256, 5

    def launch(gap, zero, stamps):
        rc = P.probe_mfma_power_launch(rnd.data_ptr(), sink.data_ptr(), stamps.data_ptr(), GRID, ITEMS, gap, zero,
                                       torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    res = {"device": torch.cuda.get_device_name(0), "precondition_s": args.pre, "matrix_instructions_per_launch": GRID * 4 * ITEMS * 960, "flops_per_launch": FLOPS,
           "alone": [], "in_step": []}
    st1 = torch.zeros(GRID, 4, dtype=torch.int64, device=dev)
    for gap, zero in ((0, 0), (0, 1), (4, 0), (8, 0), (16, 0), (32, 0), (0, 0)):
        for _ in range(5):
            launch(gap, zero, st1)
        torch.cuda.synchronize()
        import time as _t
        tp = _t.perf_counter()
        while _t.perf_counter() - tp < args.pre:
            for _ in range(50):
                launch(gap, zero, st1)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            launch(gap, zero, st1)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        r = {"gap_cycles": gap, "zero_operands": bool(zero), "pipe_duty": 32.0 / (32 + gap), "us_per_launch_events": us,
             "frac_of_2.5PF": FLOPS / (us * 1e-6) / PEAK}
        r.update(stamps_stats(st1))              # (the last launch's stamps)
        res["alone"].append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)

    # ---- inside the step: rep build + K/V pre-pass of the headline workload in front of every probe launch ----
    import bench
    from gta_amd import native, plan, synth
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, B = bench.WORKLOADS["ms-enc"]
    qm, km, vm, ex, ak, cross = synth.attention_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, seed=1234)
    q, k, v = (synth.as_projection(t, torch.bfloat16, dev) for t in (qm, km, vm))
    exd = {kk: vv.to(dev).contiguous() for kk, vv in ex.items()}
    tc = torch.tensor([0.01], device=dev)
    reps = plan.RepPlan(B, Nk, Pk, so3, so2, device=dev)
    fwd = plan.ForwardPlan(q, k, v, f_dims, so3_degree=so3, Nq=Nq, Nk=Nk)

    def prefix():
        vk, ck = reps(exd["input_transforms"], exd["input_coord"])
        fwd(q, k, v, vk, vk, ck, ck, tc, flags_extra=native.FLAG_PREP_ONLY)
        return vk, ck

    for label, gap, zero in (("real attention kernel (reference: the product step)", None, 0), ("probe", 0, 0), ("probe", 16, 0), ("probe", 0, 1),
                             ("probe", 0, 0)):
        stamps = [torch.zeros(GRID, 4, dtype=torch.int64, device=dev) for _ in range(args.steps)]
        for it in range(8):
            vk, ck = prefix()
            if gap is None:
                fwd(q, k, v, vk, vk, ck, ck, tc, flags_extra=native.FLAG_KV_READY)
            else:
                launch(gap, zero, stamps[0])
        torch.cuda.synchronize()
        import time as _t
        tp = _t.perf_counter()
        while _t.perf_counter() - tp < args.pre:
            for it in range(50):
                vk, ck = prefix()
                if gap is None:
                    fwd(q, k, v, vk, vk, ck, ck, tc, flags_extra=native.FLAG_KV_READY)
                else:
                    launch(gap, zero, stamps[0])
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(args.steps):
            vk, ck = prefix()
            if gap is None:
                fwd(q, k, v, vk, vk, ck, ck, tc, flags_extra=native.FLAG_KV_READY)
            else:
                launch(gap, zero, stamps[it])
        e1.record()
        torch.cuda.synchronize()
        step_us = e0.elapsed_time(e1) / args.steps * 1e3
        r = {"what": label, "gap_cycles": gap, "zero_operands": bool(zero), "step_us": step_us}
        if gap is not None:
            ss = [stamps_stats(s) for s in stamps]
            for key in ("span_us", "sclk_mhz", "launch_cycles"):
                r[key] = sum(s[key] for s in ss) / len(ss)
            r["frac_of_2.5PF"] = FLOPS / (r["span_us"] * 1e-6) / PEAK
        res["in_step"].append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
