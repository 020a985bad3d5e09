#!/usr/bin/env python3
"""First-contact diagnostics on a real MI355X (test infrastructure; prints, never asserts).

Runs the hardware probes, every fused-eligible golden fixture (fp32 and bf16, DMA and VGPR
staging), oracle comparisons at the CLEVR-TR / MSN shapes and a timing sweep, and writes
gpurun_out/gpu_check.json.  Usage: python tools/gpu_check.py [--quick]
"""
import ctypes
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _golden as G          # noqa: E402
from tests import _hip_cases as C       # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
report = {}


def section(name):
    print(f"\n=== {name} ===", flush=True)


def probes():
    section("probes")
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "probes", "libprobe.so"))
    lib.probe_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    res = {}
    for stride in (16, 96, 64):
        tr = torch.zeros(64, 4, dtype=torch.int16, device="cuda")
        src = torch.arange(256, dtype=torch.int32, device="cuda")
        dout = torch.zeros(512, dtype=torch.int32, device="cuda")
        rc = lib.probe_run(tr.data_ptr(), stride, src.data_ptr(), dout.data_ptr())
        tr = tr.cpu().numpy()
        # decode: element index e -> (row = e // stride, col = e % stride)
        rows, cols = tr // stride, tr % stride
        res[f"tr_stride{stride}"] = {"rc": rc, "lane0_3": tr[:4].tolist(), "lane16_17": tr[16:18].tolist(),
                                     "lane32": tr[32].tolist(), "lane48": tr[48].tolist()}
        # expected semantics: lane i of group g gets column (i&15) of the 4 rows g*4..g*4+3
        ok = True
        for l in range(64):
            g, i = l >> 4, l & 15
            for j in range(4):
                if not (rows[l, j] == g * 4 + j and cols[l, j] == i):
                    ok = False
        res[f"tr_stride{stride}"]["matches_expected_semantics"] = ok
        print(f"tr_b16 stride {stride}: rc={rc} expected-semantics={ok} lane0={tr[0].tolist()} lane1={tr[1].tolist()} "
              f"lane17={tr[17].tolist()} lane33={tr[33].tolist()}")
        d = dout.cpu().numpy().astype(np.int64) & 0xffffffff
        exp = np.full(512, 0xdeadbeef, dtype=np.int64)
        for lane in range(64):
            exp[64 + lane * 4:64 + lane * 4 + 4] = np.arange(4) + (63 - lane) * 4
        res["dma_linear_dest"] = bool((d == exp).all())
    print("LDS-DMA: dest = uniform base + lane*16 :", res["dma_linear_dest"])
    report["probes"] = res


def goldens():
    section("golden fixtures through the C ABI")
    rows = []
    for case in G.list_cases("op_"):
        _, meta = G.load("op_" + case)
        if not C.FUSED_OK(meta):
            print(f"{case:16s} skipped (not fused-eligible: {meta['f_dims']}, euclid={meta['euclid']})")
            continue
        for dtype in (torch.float32, torch.bfloat16):
            for mode in ("fused", "vgpr", "prepass"):
                for builder in ("packed", "hip"):
                    dma = mode != "vgpr"
                    if builder == "hip" and mode != "prepass":
                        continue
                    tag = f"{case}/{str(dtype)[6:]}/{mode}/{builder}"
                    try:
                        got, ref, _ = C.golden_forward(case, dtype, dma, builder, kv_mode="prepass" if mode == "prepass" else "fused")
                        st = C.err_stats(got, ref)
                        rows.append(dict(tag=tag, **st))
                        print(f"{tag:44s} max_abs={st['max_abs']:.3e} ref_max={st['ref_max']:.2f} "
                              f"rel_rms={st['rel_rms']:.3e} finite={st['finite']}", flush=True)
                    except Exception as e:  # noqa: BLE001
                        rows.append(dict(tag=tag, error=repr(e)))
                        print(f"{tag:44s} ERROR {e!r}", flush=True)
                        traceback.print_exc()
    report["goldens"] = rows


SHAPES = {
    # name: (H, Nq, Pq, Nk, Pk, f_dims, so2, so3)
    "C1": (4, 2, 64, 2, 64, {"se3": 32, "so2": 32}, 8, 0),
    "CL-enc": (6, 2, 300, 2, 300, {"se3": 32, "so2": 32}, 8, 0),
    "CL-dec": (6, 3, 853, 2, 300, {"se3": 32, "so2": 32}, 8, 0),
    "MS-enc": (8, 5, 256, 5, 256, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2),
    "MS-dec": (8, 5, 512, 5, 256, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2),
    "DT": (16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0),
}


def oracle_shapes(quick):
    section("HIP vs CPU oracle (fp32) at the BASELINE shapes, B=1")
    rows = []
    for name, (H, Nq, Pq, Nk, Pk, f_dims, so2, so3) in SHAPES.items():
        if quick and name in ("MS-dec", "CL-dec", "DT"):
            continue
        q, k, v, ex, ak, cross = C.synth_inputs(1, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, torch.float32, seed=1)
        t0 = time.time()
        ref = C.oracle_forward(q, k, v, ex, ak, cross, 0.01)
        t_or = time.time() - t0
        ref64 = C.oracle_forward(q, k, v, ex, ak, cross, 0.01, dtype=torch.float64) if name in ("C1", "CL-enc") else None
        for dtype, mode in ((torch.float32, "prepass"), (torch.bfloat16, "prepass"), (torch.bfloat16, "fused")):
            try:
                got = C.hip_forward(q, k, v, ex, ak, cross, 0.01, dtype, kv_mode=mode).float().cpu()
                # compare against the oracle fed the same (rounded) inputs
                if dtype == torch.bfloat16:
                    refd = C.oracle_forward(q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float(), ex, ak, cross, 0.01)
                else:
                    refd = ref
                st = C.err_stats(got, refd)
                rows.append(dict(shape=name, dtype=str(dtype), mode=mode, **st))
                print(f"{name:8s} {str(dtype)[6:]:9s} {mode:8s} max_abs={st['max_abs']:.3e} ref_max={st['ref_max']:.2f} "
                      f"rel_rms={st['rel_rms']:.3e} finite={st['finite']} (oracle {t_or:.2f}s)", flush=True)
            except Exception as e:  # noqa: BLE001
                rows.append(dict(shape=name, dtype=str(dtype), error=repr(e)))
                print(f"{name:8s} {dtype} ERROR {e!r}", flush=True)
                traceback.print_exc()
        if ref64 is not None:
            st = C.err_stats(ref, ref64)
            print(f"{name:8s} oracle fp32 vs fp64: max_abs={st['max_abs']:.3e}")
    report["oracle_shapes"] = rows


def timing(quick):
    section("timing (fused forward only; inputs resident; reps prebuilt)")
    import gta_amd
    rows = []
    for name, (H, Nq, Pq, Nk, Pk, f_dims, so2, so3) in SHAPES.items():
        for B in ((32,) if quick else (1, 8, 32)):
            if name == "C1" and B != 32:
                continue
            for dtype in (torch.bfloat16, torch.float32):
                for mode in ("prepass", "fused"):
                    dma = mode
                    try:
                        q, k, v, ex, ak, cross = C.synth_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, dtype, seed=2)
                        exd = {kk: vv.cuda() for kk, vv in ex.items()}
                        gta_amd.pre_compute_reps_encoder(ak, exd)
                        if cross:
                            gta_amd.pre_compute_reps_decoder(ak, exd)
                        packed = gta_amd.pack_reps(exd, f_dims)
                        tc = torch.tensor([0.01], device="cuda") if f_dims.get("se3", 0) > 0 else None
                        # packed-projection layout [B,T,H,dh] like the module produces
                        qd = q.to(dtype).cuda().permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
                        kd = k.to(dtype).cuda().permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
                        vd = v.to(dtype).cuda().permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
                        fn = lambda: gta_amd.gta_attention(qd, kd, vd, f_dims, packed, so3_degree=so3, trans_coeff=tc,
                                                           kv_mode=mode)
                        for _ in range(3):
                            fn()
                        torch.cuda.synchronize()
                        n = 10
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(n):
                            fn()
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1) / n
                        Tq, Tk, dh = Nq * Pq, Nk * Pk, sum(f_dims.values())
                        flops = 4.0 * B * H * Tq * Tk * dh
                        tf = flops / (ms * 1e-3) / 1e12
                        rows.append(dict(shape=name, B=B, dtype=str(dtype), dma=dma, ms=ms, tflops=tf,
                                         mtok_s=B * Tq / (ms * 1e-3) / 1e6))
                        print(f"{name:8s} B={B:3d} {str(dtype)[6:]:9s} {mode:8s} {ms:8.3f} ms  "
                              f"{tf:7.1f} TFLOP/s ({100 * tf / 2500:.1f}% of bf16 MFMA peak)  "
                              f"{B * Tq / (ms * 1e-3) / 1e6:8.2f} Mtok/s", flush=True)
                    except Exception as e:  # noqa: BLE001
                        rows.append(dict(shape=name, B=B, dtype=str(dtype), dma=dma, error=repr(e)))
                        print(f"{name:8s} B={B} {dtype} dma={dma} ERROR {e!r}", flush=True)
    report["timing"] = rows


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    print(torch.__version__, torch.cuda.get_device_name(0))
    for fn in (probes, goldens, lambda: oracle_shapes(quick), lambda: timing(quick)):
        try:
            fn()
        except Exception:  # noqa: BLE001
            traceback.print_exc()
    with open(os.path.join(OUT, "gpu_check.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("\nwrote gpurun_out/gpu_check.json")
