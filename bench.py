#!/usr/bin/env python3
"""bench.py -- GTA-attention throughput on MI355X (contract: one JSON line on rank 0).

A "step" is one pass of the hot path over one synthetic batch, inputs resident in HBM: build the per-view /
per-token reps from poses and patch coordinates on device (`gta_build_reps`), then ONE GTA attention forward
(`gta_attn_fwd`: K/V rep pre-pass + attention kernel) on already-projected Q/K/V.  Default workload = BASELINE.json's
metric configuration: the MSN-Hard `gta_so3` encoder attention (V=5 views of 128x128 images -> 16x16 patches/view ->
1280 tokens, d=768 = 8 heads x 96 channels, f_dims se3 48 / so3 24 / so2 24), bf16, 32 scenes per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The step goes through the C ABI with pre-planned buffers (gta_amd.plan: what a serving loop does -- no allocation, no
autograd bookkeeping per call).  The dominant kernel's duration is taken inside the timed region from HIP events
attached to its own dispatch (hipExtLaunchKernelGGL through the library's profiling hook, on the launch stream).
After the timed region the output of the full-size batch is checked against the CPU oracle on sampled scenes
(`parity_max_abs`).  `oracle/` is imported for that check and for the `cpu_baseline` leg only.

Multi-GPU: data-parallel replicas (one process per GPU, fixed per-GPU batch -> "weak"); the operator has no exchange
step, so there is no data-path collective -- only the barrier and the MAX-over-ranks of the timed region go through RCCL.
"""
import argparse
import ctypes
import glob
import json
import os
import sys
import time

# The GPU boxes run this in a container with a CPU QUOTA (cgroup cpu.max: 16 CPUs' worth per 100-ms period on the r05 boxes) under 256
# visible CPUs.  torch's default of 128 intra-op threads, spinning after every small CPU op, burns that quota in a fraction of the period and
# the WHOLE process -- the thread that launches kernels included -- is then descheduled until the period ends: 70-ms holes in the launch
# stream every 100 ms (tools/clock_ramp.py "small" shows them).  So: passive OpenMP waits, and a thread count inside the quota.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PEAK_CLOCK_MHZ = 2400.0         # the shader clock the 2.5 PFLOP/s figure assumes (MI355X_MICROARCH.md: max clock)
MFMA_FLOP_PER_CYCLE = 1024.0 * 4 * 256   # v_mfma_f32_32x32x16_bf16: 32768 flop / 32 cycles per SIMD, 4 SIMDs x 256 CUs

WORKLOADS = {
    # name: (H, Nq, Pq, Nk, Pk, f_dims, so2, so3, default per-GPU batch)
    "ms-enc": (8, 5, 256, 5, 256, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2, 32),
    "ms-dec": (8, 5, 512, 5, 256, {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, 6, 2, 32),
    "cl-enc": (6, 2, 300, 2, 300, {"se3": 32, "so2": 32}, 8, 0, 32),
    "cl-dec": (6, 3, 853, 2, 300, {"se3": 32, "so2": 32}, 8, 0, 32),
    "dit": (16, 1, 1024, 1, 1024, {"so2": 64}, 16, 0, 32),
    # not BASELINE configs, not in the default line (`--workload` only): the MSN runs without the so3 slab (runs/msn/GTA/gta: se3 48 | so2 48)
    "ms-gta-enc": (8, 5, 256, 5, 256, {"triv": 0, "se3": 48, "so2": 48}, 12, 0, 32),
    "ms-gta-dec": (8, 5, 512, 5, 256, {"triv": 0, "se3": 48, "so2": 48}, 12, 0, 32),
    "ms-se3-enc": (8, 5, 256, 5, 256, {"se3": 96}, 0, 0, 32),        # the encoder of runs/msn/GTA/gta_no2demb
}


def cpu_quota():
    """CPUs this process may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota), capped by the visible CPUs"""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


HOST_THREADS = max(1, min(8, cpu_quota()))      # torch's intra-op threads while the GPU is being timed (the launching thread needs one of the quota's CPUs)


def oracle_forward(q, k, v, ex, ak, cross, trans_coeff):
    """CPU oracle (oracle/gta_oracle.py: a restatement of the reference's PyTorch path, pinned to it by tests/golden)."""
    from oracle import gta_oracle as O
    reps = O.encoder_reps(ak, {kk: vv.float() for kk, vv in ex.items()})
    if cross:
        reps = O.decoder_reps(ak, {kk: vv.float() for kk, vv in ex.items()}, reps)
    out, _ = O.gta_attention(q.float(), k.float(), v.float(), ak["f_dims"], reps, trans_coeff, True)
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(workload, seed):
    """The CPU oracle on a bounded sample of the same workload, host cores.  Reported beside the GPU number; baseline only."""
    from gta_amd import synth
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, _ = WORKLOADS[workload]
    Bs = 2
    q, k, v, ex, ak, cross = synth.attention_inputs(Bs, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, seed=seed)
    ncpu = os.cpu_count() or 1
    quota = cpu_quota()
    best = None
    t_end = time.time() + 25.0
    # intra-op threading of many small einsums stops scaling early: try a few counts (inside the container's CPU quota), keep the best
    for nthreads in sorted({min(quota, n) for n in (8, 16, 32, 64)}):
        if time.time() > t_end:
            break
        torch.set_num_threads(nthreads)
        for _ in range(2):                                    # two warm-ups
            oracle_forward(q, k, v, ex, ak, cross, 0.01)
        times = []
        while len(times) < 5 and time.time() < t_end:
            t0 = time.perf_counter()
            oracle_forward(q, k, v, ex, ak, cross, 0.01)
            times.append(time.perf_counter() - t0)
        if times:
            times.sort()
            med_n = times[len(times) // 2]
            if best is None or med_n < best[0]:
                best = (med_n, nthreads, len(times))
    med, cores, nruns = best
    torch.set_num_threads(HOST_THREADS)
    return {"value": Bs * Nq * Pq / med / 1e6, "unit": "Mtokens/s", "cores": cores, "host_cpus": ncpu, "host_cpu_quota": quota,
            "cpu_model": cpu_model(), "kind": "port", "warmups": 2,
            "sample": f"oracle/gta_oracle.py fp32 (rep build + attention), B={Bs} scenes of the same workload, "
                      f"median of {nruns} runs, {med * 1e3:.1f} ms each"}


def parity_check(out, masters, ak, cross, scenes):
    """Output of the timed configuration (full batch) against the oracle on a few scenes, all heads."""
    q, k, v, ex = masters
    idx = torch.tensor(scenes)
    torch.set_num_threads(max(1, min(16, cpu_quota())))
    try:
        ref = oracle_forward(q[idx], k[idx], v[idx], {kk: vv[idx] for kk, vv in ex.items()}, ak, cross, 0.01)
    finally:
        torch.set_num_threads(HOST_THREADS)
    got = out[idx.to(out.device)].float().cpu()
    diff = (got - ref).abs()
    return {"scenes": list(scenes), "parity_max_abs": float(diff.max()), "ref_max_abs": float(ref.abs().max()),
            "rel_rms": float(((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))}


def latest_traffic(workload, B, dtype, kernel):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC measurement of this workload AND this kernel
    (a counter pass cannot share a run with the timed region; a measurement of another kernel is not quoted)."""
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{workload}.json"))
    for f in sorted(files, reverse=True):
        try:
            t = json.load(open(f))
            if (t["workload"], t["batch"], t["dtype"]) == (workload, B, dtype) and t.get("kernel", "gta_fwd2_kernel") == kernel:
                return t["bytes_per_launch"], t.get("mfma_busy_sq"), os.path.relpath(f, ROOT) + " (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)"
        except (OSError, KeyError, ValueError):
            pass
    return None, None, None


def kernel_clock(prof):
    """[n_items, 8] stamps of one launch (include/gta_hip.h: gta_debug_profile_next_attention_kernel) -> (shader cycles of the launch, granted
    shader clock in MHz): span of the 100-MHz stamps x the clock the items' own cycle counts give."""
    P = prof.cpu().double()
    real = P[:, 6] - P[:, 5]
    ok = real > 0
    if not bool(ok.any()):
        return None, None
    mhz = float(((P[:, 4] - P[:, 0])[ok] / real[ok]).mean()) * 100.0
    span_us = float(P[ok][:, 6].max() - P[ok][:, 5].min()) / 100.0
    return span_us * mhz, mhz


def precondition(fn, seconds, chunk=20):
    """Run fn back to back for `seconds` (untimed): the GPU settles on its sustained clock after about a second of load (tools/clock_ramp.py);
    every timed figure of this file is taken behind such a phase of ITS OWN work.  Returns the number of calls."""
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(chunk):
            fn()
        n += chunk
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    return n


class PlannedStep:
    """One workload's step through the planned C-ABI path: rep build(s) + ONE gta_attn_fwd (K/V pre-pass + attention kernel), inputs resident
    in HBM; `step(i)` of a sampled index i also attaches dispatch events and per-item stamps to the attention kernel's own launch."""

    def __init__(self, workload, B, dtype_name, device, L, seed, steps, kernel_samples, flags=0, time_kernel=True):
        import gta_amd
        from gta_amd import plan, synth
        H, Nq, Pq, Nk, Pk, f_dims, so2, so3, _ = WORKLOADS[workload]
        self.L, self.B, self.H, self.f_dims = L, B, H, f_dims
        self.dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
        qm, km, vm, ex, ak, cross = synth.attention_inputs(B, H, Nq, Pq, Nk, Pk, f_dims, so2, so3, seed=seed)
        self.masters, self.ak, self.cross = (qm, km, vm, ex), ak, cross
        # Q/K/V as the module's packed projection leaves them: [B, T, H, dh] in memory, viewed [B, H, T, dh]
        self.q, self.k, self.v = (synth.as_projection(t, self.dtype, device) for t in (qm, km, vm))
        self.exd = {kk: vv.to(device).contiguous() for kk, vv in ex.items()}
        self.tc = torch.tensor([0.01], device=device) if f_dims.get("se3", 0) > 0 else None
        self.Tq, self.Tk, self.dh = Nq * Pq, Nk * Pk, sum(f_dims.values())
        need_view = f_dims.get("se3", 0) > 0 or f_dims.get("so3", 0) > 0
        self.so3_deg = so3 if f_dims.get("so3", 0) > 0 else 0
        # ---- the planned step: rep build(s) + one gta_attn_fwd ----
        self.reps_k = plan.RepPlan(B, Nk, Pk, self.so3_deg, so2, device=device) if (need_view and f_dims.get("so2", 0) > 0) else None
        self.reps_q = plan.RepPlan(B, Nq, Pq, self.so3_deg, so2, device=device) if (self.reps_k is not None and cross) else None
        self.fwd = plan.ForwardPlan(self.q, self.k, self.v, f_dims, so3_degree=self.so3_deg, Nq=Nq if need_view else 1,
                                    Nk=Nk if need_view else 1, flags=flags)
        n_it, rows_it = ctypes.c_int32(0), ctypes.c_int32(0)
        self.kname = (L.gta_debug_attention_kernel(ctypes.byref(self.fwd.desc), ctypes.byref(n_it), ctypes.byref(rows_it)) or b"").decode()
        self.rows_it = rows_it.value
        time_kernel = time_kernel and self.kname != "gta_fwd_kernel"       # (the single-kernel plan has no separate attention launch to bracket)
        self.time_kernel = time_kernel
        n_samp = steps if kernel_samples <= 0 else min(kernel_samples, steps)
        stride = max(1, steps // n_samp)
        self.sampled = {i: len(range(stride // 2, i, stride)) for i in range(stride // 2, steps, stride)} if time_kernel else {}   # step -> event slot
        self.ev = [(L.gta_debug_event_create(), L.gta_debug_event_create()) for _ in self.sampled]
        self.n_samples = len(self.ev)
        # the sampled launches also leave per-item start / end stamps (shader cycles + 100-MHz clock): kernel cycles and granted clock
        self.profs = [torch.zeros(max(n_it.value, 1), 8, dtype=torch.int64, device=device) for _ in self.sampled]
        self._gta = gta_amd

    def build_reps(self):
        exd, cross = self.exd, self.cross
        if self.reps_k is not None:
            vk, ck = self.reps_k(exd["input_transforms"], exd["input_coord"])
            vq, cq = self.reps_q(exd["target_transforms"], exd["target_coord"]) if cross else (vk, ck)
            return vq, vk, cq, ck
        # layouts without a per-view part (or without so2): the general builders of gta_amd.reps
        e2 = dict(exd)
        self._gta.pre_compute_reps_encoder(self.ak, e2)
        if cross:
            self._gta.pre_compute_reps_decoder(self.ak, e2)
        pk = self._gta.pack_reps(e2, self.f_dims)
        return pk.get("vrep_q"), pk.get("vrep_k"), pk.get("cs_q"), pk.get("cs_k")

    def step(self, i=None):
        L = self.L
        vq, vk, cq, ck = self.build_reps()                 # reps are rebuilt every step (timed)
        if i in self.sampled:
            e = self.ev[self.sampled[i]]
            L.gta_debug_time_next_attention_kernel(ctypes.c_void_p(e[0]), ctypes.c_void_p(e[1]))
            pb = self.profs[self.sampled[i]]
            L.gta_debug_profile_next_attention_kernel(ctypes.c_void_p(pb.data_ptr()), pb.shape[0])      # (one-shot: this launch only)
        return self.fwd(self.q, self.k, self.v, vq, vk, cq, ck, self.tc)

    def kernel_times(self):
        """(mean attention-kernel ms by the events of its own dispatch, mean shader cycles per launch, granted clock in MHz) over the sampled steps"""
        if not self.ev:
            return None, None, None
        L = self.L
        ks = [L.gta_debug_event_elapsed_ms(ctypes.c_void_p(a), ctypes.c_void_p(b)) for a, b in self.ev]
        kern_ms = sum(ks) / len(ks)                         # attention kernel only (events of its own dispatch)
        cyc = [kernel_clock(p_) for p_ in self.profs]
        cyc = [c for c in cyc if c[0]]
        kern_cycles = sum(c[0] for c in cyc) / len(cyc) if cyc else None
        sclk_mhz = sum(c[1] for c in cyc) / len(cyc) if cyc else None
        return kern_ms, kern_cycles, sclk_mhz

    def release_events(self):
        for a, b in self.ev:
            self.L.gta_debug_event_destroy(ctypes.c_void_p(a)); self.L.gta_debug_event_destroy(ctypes.c_void_p(b))
        self.ev = []

    def flops(self):
        return 4.0 * self.B * self.H * self.Tq * self.Tk * self.dh      # QK^T + PV, 2 flop/MAC


def workload_leg(name, dtype_name, device, L, seed, steps=100, warmup=10, kernel_samples=6, bwd_steps=5, precise=False, precondition_s=1.2):
    """One of the OTHER BASELINE workloads, measured exactly as the headline (same planned step, same dispatch events and stamps), in well
    under a second of GPU time -> the entry of the line's `workloads` object.  Never part of `value`."""
    import gta_amd
    from gta_amd import native
    B = WORKLOADS[name][8]
    # precise = the fp32-faithful mode (fp32 inputs, split-bf16 operands, three MFMAs per product; at dh <= 64 on the two-stage plan)
    ps = PlannedStep(name, B, dtype_name, device, L, seed=seed, steps=steps, kernel_samples=kernel_samples,
                     flags=native.FLAG_FP32_PRODUCTS if precise else 0, time_kernel=True)
    import gc
    gc.collect()                                           # (before the GPU is warmed: a collection is tens of milliseconds of idle GPU)
    gc.disable()
    precondition(ps.step, precondition_s)
    for _ in range(warmup):
        ps.step()
    torch.cuda.synchronize()
    # (stream events around the 100 steps, three times, the best one counts: the short workloads' timed region is a few milliseconds, and on
    #  the boxes of r05 one region in ten ran 3-8x long with the attention kernel's own time unchanged -- the host behind with its launches,
    #  right after the previous leg's oracle check had the CPU's cores; all three are reported)
    regions = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            ps.step(i if rep == 0 else None)               # (the dispatch events and stamps ride on the first region's sampled steps)
        e1.record()
        torch.cuda.synchronize()
        regions.append(e0.elapsed_time(e1) / steps)
    gc.enable()
    ms = min(regions)
    ms_med = sorted(regions)[len(regions) // 2]            # (the median of the three regions rides beside the best one: ADVICE r05)
    kern_ms, cyc, mhz = ps.kernel_times()
    ps.release_events()
    fl = ps.flops()
    out = {"value": B * ps.Tq / (ms * 1e-3) / 1e6, "value_median": B * ps.Tq / (ms_med * 1e-3) / 1e6, "unit": "Mtokens/s", "ms_per_step": ms,
           "ms_per_step_median": ms_med, "steps": steps, "warmup": warmup, "batch": B, "dtype": dtype_name,
           "mode": "fp32-faithful products (GTA_FLAG_FP32_PRODUCTS)" if precise else "default (bf16 products, fp32 accumulation)",
           "kernel": ps.kname, "kernel_ms": kern_ms, "frac": (fl / (kern_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if kern_ms else None,
           "step_frac": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "kernel_cycles": cyc, "sclk_mhz": mhz, "ms_per_step_regions": regions,
           "mfma_busy": (fl / MFMA_FLOP_PER_CYCLE / cyc) if cyc else None, "algorithmic_flops": fl,
           "shape": {"H": ps.H, "Tq": ps.Tq, "Tk": ps.Tk, "dh": ps.dh}}
    if not precise:
        # HBM bytes per launch of this workload's attention kernel from its committed PMC pass (profiles/rNN/pmc_<workload>.json), as `roofline.traffic`
        tr, busy_sq, src = latest_traffic(name, B, dtype_name, ps.kname)
        out.update({"traffic": tr, "traffic_source": src, "mfma_busy_sq": busy_sq,
                    "algorithmic_bytes": (2 * ps.Tq + 2 * ps.Tk) * ps.H * ps.dh * (2 if dtype_name == "bf16" else 4) * B})
    # the full-batch output against the oracle on one scene (the GPU tests hold the full parity matrix of this workload)
    out["parity"] = parity_check(ps.fwd.out, ps.masters, ps.ak, ps.cross, [B - 1])
    if bwd_steps > 0:
        qg, kg, vg = (t.detach().clone().requires_grad_() for t in (ps.q, ps.k, ps.v))
        tcg = ps.tc.detach().clone().requires_grad_() if ps.tc is not None else None
        e2 = dict(ps.exd)
        gta_amd.pre_compute_reps_encoder(ps.ak, e2)
        if ps.cross:
            gta_amd.pre_compute_reps_decoder(ps.ak, e2)
        packed = gta_amd.pack_reps(e2, ps.f_dims)
        w = torch.randn_like(ps.q)

        def train_step():
            o = gta_amd.gta_attention(qg, kg, vg, ps.f_dims, packed, so3_degree=e2.get("gta_so3_degree", 0), trans_coeff=tcg, precise=precise)
            o.backward(w)
            qg.grad = kg.grad = vg.grad = None
        precondition(train_step, min(0.6, precondition_s), 5)
        for _ in range(3):
            train_step()
        torch.cuda.synchronize()
        fb = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(bwd_steps):
                train_step()
            e1.record()
            torch.cuda.synchronize()
            fb.append(e0.elapsed_time(e1) / bwd_steps)
        out["fwd_bwd_ms"] = min(fb)
        out["fwd_bwd_ms_median"] = sorted(fb)[len(fb) // 2]
        out["fwd_bwd_ms_regions"] = fb
    return out


def render_leg(device, B=4, chunk=20480, reps=3, precondition_s=0.6):
    """SURVEY 8 f4 at the CLEVR-TR size: full-image decode of one 240 x 320 target view per scene (76 800 query rays, trainer.py:137-181) through the
    CLEVR-TR model (runs/clevrtr/GTA/gta/config.yaml: 600 scene tokens, dh = 64), chunked like the reference (max_num_rays = num_points * batch_size / B
    = 20 480 at B = 4, trainer.py:155-157), per-layer K'/V' images cached across the chunks; bf16 autocast, random-init weights.  Never part of `value`."""
    from gta_amd import srt
    torch.manual_seed(4321)
    model = srt.TransformingSRT(srt.clevrtr_gta_cfg(dropout=0.0)).to(device).eval()
    data = srt.synthetic_batch(B, n_in=2, n_tgt=1, image=(120, 160), points_per_view=8, device=device, seed=17)
    h, w = 240, 320
    rays = torch.nn.functional.normalize(torch.randn(B, h, w, 3, device=device), dim=-1)
    cam = torch.randn(B, 3, device=device)
    extras = {"input_transforms": data["input_transforms"], "input_coord": data["input_coord"], "target_transforms": data["target_transforms"][:, :1]}
    out = {"config": f"CLEVR-TR TransformingSRT decoder (2 cross-attention blocks, 6 heads x 64, Tk = 600) + render MLP, B={B} scenes x one 240x320 view, "
                     f"{chunk}-ray chunks, K'/V' cache reused, bf16 autocast", "batch": B, "rays_per_view": h * w, "chunk": chunk}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        z, extras = model.encoder(data["input_images"], data["input_camera_pos"], data["input_rays"], extras)

        def run(reuse):
            return srt.render_image(model, z, cam, rays, extras, max_num_rays=chunk, reuse_kv=reuse)[0]
        for reuse, key in ((True, "ms_per_image_batch"), (False, "ms_per_image_batch_no_cache")):
            precondition(lambda: run(reuse), precondition_s, 2)
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                img = run(reuse)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[key] = min(ts)
            out[key + "_median"] = sorted(ts)[len(ts) // 2]
        out["finite"] = bool(torch.isfinite(img).all())
    out["value"] = B * h * w / (out["ms_per_image_batch"] * 1e-3) / 1e6
    out["value_median"] = B * h * w / (out["ms_per_image_batch_median"] * 1e-3) / 1e6
    out["unit"] = "Mrays/s"
    return out


def model_train_leg(args, dist, world, rank, local_rank, device, model, loss_of):
    """K optimizer steps of ``model`` (sub-modules ``encoder`` / ``decoder``) under the reference's data-parallel structure
    (train.py:182-188: each in its OWN DistributedDataParallel -> two bucketed gradient all-reduce streams per step, RCCL over xGMI on
    the GPUs, gloo in --dry-run), timed between barriers with the MAX over ranks; then the bucket timeline on two EXTRA steps (the
    Python hook replaces DDP's C++ all-reduce, so it stays out of the timed steps).  -> the `srt_train` object of the JSON line."""
    from gta_amd import ddp
    cuda = device.type == "cuda"
    if dist is not None:
        model, _ = ddp.wrap_srt_ddp(model, local_rank, logged=False)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)

    def sync():
        if cuda:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def model_step():
        opt.zero_grad(set_to_none=True)
        loss = loss_of(model)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        model_step()
    sync()
    tm0 = time.perf_counter()
    for _ in range(args.model_train_steps):
        last = model_step()
    sync()
    dtm = time.perf_counter() - tm0
    if dist is not None:
        tt = torch.tensor([dtm], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dtm = float(tt.item())
    n_ = max(world, 1)
    ms = dtm / args.model_train_steps * 1e3
    out = {"ms_per_step": ms, "steps": args.model_train_steps, "scenes_per_s": n_ * args.model_batch / (ms * 1e-3),
           "loss": float(last.item()), "grad_allreduce": None,
           "ddp": "two DistributedDataParallel instances (encoder, decoder) as train.py:182-188" if dist is not None else None,
           "note": "whole-model optimizer step on synthetic batches (SURVEY 8 f2); not part of `value`"}
    if dist is not None:
        logs = []
        for sub in (model.encoder, model.decoder):
            lg = ddp.BucketLog()
            sub.register_comm_hook(None, lg.hook)
            logs.append(lg)
        for _ in range(2):
            model_step()
        sync()
        out["grad_allreduce"] = {nm: lg.summary(2) for nm, lg in zip(("encoder", "decoder"), logs)}
    return out


def dry_run(args):
    """The multi-process control flow of main() on CPU: rendezvous from the launcher's environment, barrier, timed loop,
    barrier, MAX over ranks, per-rank times gathered, one JSON line on rank 0.  No kernel runs; `value` is meaningless."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        dist.init_process_group("gloo")
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, Bdef = WORKLOADS[args.workload]
    B = args.batch or Bdef
    x = torch.zeros(8)
    for _ in range(args.warmup):
        x += 1
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x += 1
        time.sleep(0.001 * (1 + rank))
    mine = time.perf_counter() - t0                        # this rank's own loop (before it waits for the others)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [mine / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, torch.tensor([mine / args.steps * 1e3], dtype=torch.float64))
        per_rank = [float(u.item()) for u in g]
    # the whole-model leg's control flow (two DDP instances, timed optimizer steps between barriers, MAX over ranks, bucket timeline)
    # on a stand-in model with the same encoder / decoder structure: the same function the GPU run calls
    srt_train = None
    if args.model_train_steps < 0:
        args.model_train_steps = 5 if world > 1 else 0
    if args.model_train_steps > 0:
        torch.manual_seed(7)
        model = torch.nn.Module()
        model.encoder = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 16))
        model.decoder = torch.nn.Sequential(torch.nn.Linear(16, 8))
        xb = torch.randn(args.model_batch, 16, generator=torch.Generator().manual_seed(99 + rank))
        srt_train = model_train_leg(args, dist, world, rank, 0, torch.device("cpu"), model,
                                    lambda m: m.decoder(m.encoder(xb)).square().mean())
    if rank == 0:
        n = max(world, 1)
        print(json.dumps({"metric": "GTA-attn Mtokens/s (V=5,H=W=128,d=768)", "value": n * B * Nq * Pq * args.steps / elapsed / 1e6,
                          "srt_train": srt_train,
                          "unit": "Mtokens/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "per_rank_ms_per_step": per_rank,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                          "data": "synthetic", "dry_run": True,
                          "config": {"workload": args.workload, "global_batch": n * B, "parallelism": f"dp{n}"}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def block_layer_leg(B, steps, device, precondition_s=0.6):
    """one MSN encoder layer (d = 768 = 8 x 96, mlp 1536, 5 views x 256 tokens, bf16 autocast, dropout off) run
    as the fused block and module by module -- microseconds per layer, forward and forward + backward."""
    import gta_amd
    from gta_amd import layers, synth
    torch.manual_seed(0)
    ak = {"f_dims": {"triv": 0, "se3": 48, "so3": 24, "so2": 24}, "so2": 6, "so3": 2, "max_freq_h": 1, "max_freq_w": 1}
    tr = gta_amd.Transformer(768, 1, 8, 96, 1536, 0.0, True, None, False, {"method": {"name": "gta", "args": ak}}).to(device)
    gen = torch.Generator().manual_seed(1)
    ex = {"input_transforms": synth.random_extrinsics(B, 5, gen).to(device), "input_coord": torch.rand(B, 5, 16, 16, 2, generator=gen).to(device)}
    gta_amd.pre_compute_reps_encoder(ak, ex)
    x0 = torch.randn(B, 1280, 768, device=device)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            tr(x0, None, ex)

    def fwd_bwd():
        x = x0.requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = tr(x, None, ex)
        y.backward(x0)
        for p in tr.parameters():
            p.grad = None
        x0.grad = None

    def timeit(fn):
        precondition(fn, precondition_s, 5)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps * 1e3

    res = {}
    try:
        for label, on in (("modules", False), ("fused", True)):
            tr.fused_blocks = on
            res[label] = {"forward_us": timeit(fwd), "forward_backward_us": timeit(fwd_bwd)}
    finally:
        tr.fused_blocks = True
    res["config"] = f"one pre-LN layer of the MSN gta_so3 encoder: d=768 (8 x 96), mlp 1536, B={B}, 1280 tokens, bf16 autocast"
    res["note"] = "fused = libgta_block.so (DESIGN.md section 9); modules = nn.LayerNorm / nn.Linear / nn.GELU + autograd; not part of `value`"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="ms-enc", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the full-batch output")
    ap.add_argument("--model-train-steps", type=int, default=-1,
                    help="also time K optimizer steps of the whole MSN gta_so3 TransformingSRT (gta_amd.srt) on synthetic "
                         "multi-view batches, DDP over the ranks (RCCL's bucketed gradient all-reduce: the one collective of the "
                         "north star); reported as `srt_train`, not part of `value`.  Default: 5 when launched on more than one "
                         "GPU (so the driver's `--gpus N --steps K --warmup W` line carries the collective), 0 on one GPU")
    ap.add_argument("--model-batch", type=int, default=8, help="per-GPU scenes for --model-train-steps")
    ap.add_argument("--train-steps", type=int, default=10,
                    help="extra (untimed-for-`value`) forward+backward steps reported as fwd_bwd_* fields; 0 = skip")
    ap.add_argument("--kernel-samples", type=int, default=6,
                    help="how many of the K timed steps carry the dispatch events that time the attention kernel (evenly "
                         "spaced over the timed region).  The events cost ~7 us of launch gap in front of the kernel they "
                         "bracket (profiles/r02/final_step_gaps.txt), so bracketing every step would slow the metric it "
                         "sits in; 0 = every step")
    ap.add_argument("--block-steps", type=int, default=5,
                    help="also time one whole pre-LN layer around the operator (SURVEY 8 f1: the fused block of libgta_block.so "
                         "against the module-by-module path), forward and forward+backward; reported as `block_layer`, not "
                         "part of `value`; 0 = skip")
    ap.add_argument("--workloads", default="auto",
                    help="comma-separated list of OTHER BASELINE workloads to measure after the headline's timed region (100 steps each, same "
                         "planned step, dispatch events and per-item stamps; under a second of GPU time each) -> the `workloads` object of the "
                         "JSON line; never part of `value`.  auto = ms-dec,cl-enc,cl-dec,dit when the headline workload is ms-enc on bf16 and "
                         "the run has one rank, none otherwise; none = skip")
    ap.add_argument("--precondition-s", dest="precondition_s", type=float, default=1.2,
                    help="seconds of the SAME step run back to back, untimed, BEFORE the W warmup steps and the timed region.  A process that has "
                         "just built its inputs starts on an idle GPU; MI355X then grants the attention kernel 1.30-1.50 GHz for the first tens of "
                         "milliseconds and settles at 1.75-1.77 GHz after about a second of load, where it stays (tools/clock_ramp.py, "
                         "profiles/r05/raw/clock_ramp.txt: 0.197 ms/step from t = 1 s to t = 6 s).  `--steps 20 --warmup 5` is 6 ms of GPU work: "
                         "without this phase `value` is a number about the clock ramp, not about the kernels.  The line reports both: `value` "
                         "(sustained) and `cold_start` (W warmup + K steps from idle, what rounds 1-4 reported).  0 = no preconditioning, "
                         "value == the cold number")
    ap.add_argument("--kv-mode", dest="kv_mode", default="prepass", choices=["prepass", "fused", "prepass_rows32", "prepass_fwd2", "prepass_item_cxx", "prepass_bwd_keys32", "prepass_bwd_keys64", "prepass_bwd_split", "prepass_bwd_keys64_split"],
                    help="execution plan of gta_attn_fwd (see include/gta_hip.h); prepass_rows32 = GTA_FLAG_ROWS32: the 32-rows-per-wave "
                         "attention kernel where the 64-rows one would run; prepass_item_cxx = GTA_FLAG_ITEM_CXX: the 64-rows kernel with its "
                         "compiler-scheduled item prologue / epilogue where the generated item stream would run (A/B)")
    ap.add_argument("--precise", action="store_true",
                    help="fp32-faithful forward for fp32 inputs (GTA_FLAG_FP32_PRODUCTS: split-bf16 operands, three MFMAs per product, "
                         "single-kernel plan; the reference's mixed_prec: False configs); needs --dtype f32; its fwd_bwd leg: fp32 rho kernels, "
                         "split-bf16 plain forward, exact-fp32 backward (gta_plain32.hip)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a GPU: same launch contract, rendezvous (gloo), barriers, MAX-over-ranks and "
                         "JSON line, the step itself replaced by a host no-op (tests/test_ddp_gloo.py runs this at world size 2)")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    import gta_amd
    from gta_amd import native, plan, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    global HOST_THREADS
    HOST_THREADS = max(1, min(8, cpu_quota() // max(world, 1)))      # (the ranks of one node share the container's quota)
    torch.set_num_threads(HOST_THREADS)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    L = native.lib()   # fail loudly if the HIP library is missing
    if args.model_train_steps < 0:
        args.model_train_steps = 5 if world > 1 else 0

    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, Bdef = WORKLOADS[args.workload]
    B = args.batch or Bdef
    if args.precise:
        if args.dtype != "f32":
            raise SystemExit("--precise is the fp32-faithful mode: use --dtype f32")
        # (r05: at dh <= 64 the mode runs the two-stage plan -- split hi / lo images + three MFMAs per product in the 32-row kernel -- unless
        #  --kv-mode fused asks for the single-kernel plan; its fwd_bwd leg runs rho in fp32 + the exact-fp32 backward of gta_plain32.hip)
    fused = args.kv_mode == "fused"
    ps = PlannedStep(args.workload, B, args.dtype, device, L, seed=1234 + rank, steps=args.steps, kernel_samples=args.kernel_samples,
                     flags=(native.FLAG_FP32_PRODUCTS if args.precise else 0) | (native.FLAG_FUSED_KV if fused
                     else native.FLAG_ROWS32 if args.kv_mode == "prepass_rows32"
                     else (native.FLAG_ROWS32 | native.FLAG_FWD2_GENERIC) if args.kv_mode == "prepass_fwd2"
                     else native.FLAG_ITEM_CXX if args.kv_mode == "prepass_item_cxx" else 0), time_kernel=not fused)
    step, fwd, q, k, v, exd, tc, ak, cross, kname, rows_it = ps.step, ps.fwd, ps.q, ps.k, ps.v, ps.exd, ps.tc, ps.ak, ps.cross, ps.kname, ps.rows_it
    qm, km, vm, ex = ps.masters
    Tq, Tk, dh = ps.Tq, ps.Tk, ps.dh
    dtype = ps.dtype

    # ---- cold start (reported beside `value`): W warmup + K steps from the idle GPU this process starts on -- rounds 1-4's `value` ----
    cold_start, precond = None, None
    if args.precondition_s > 0:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for i in range(args.steps):
            step(i)                                         # (the same sampled steps carry dispatch events and stamps as in the timed region below)
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc0) / args.steps * 1e3
        c_kern_ms, c_cycles, c_mhz = ps.kernel_times()      # read out now: the timed region re-records the same events and stamp buffers
        c_flops = 4.0 * B * H * Tq * Tk * dh
        cold_start = {"value": B * Tq / (cold_ms * 1e-3) / 1e6, "ms_per_step": cold_ms, "steps": args.steps, "warmup": args.warmup,
                      "kernel_ms": c_kern_ms, "frac": (c_flops / (c_kern_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if c_kern_ms else None,
                      "kernel_cycles": c_cycles, "sclk_mhz": c_mhz,
                      "note": "this rank, W warmup + K steps right after the process built its inputs (idle GPU): the protocol of rounds 1-4"}
        # ---- preconditioning: the same step, untimed, until the part has settled on its sustained clock ----
        tp0 = time.perf_counter()
        n_pre = precondition(step, args.precondition_s, 50)      # (synchronised every 50 steps: the host enqueues a step in 26 us, the GPU runs it in 200)
        precond = {"seconds": time.perf_counter() - tp0, "steps": n_pre,
                   "note": "untimed steps in front of the W warmup steps: the GPU's clock settles after ~1 s of load (tools/clock_ramp.py)"}
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_host = time.perf_counter() - t0                      # (host side of the loop: how far ahead of the GPU it runs)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0                        # this rank's own K steps (before it waits for the others)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [mine / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, torch.tensor([mine / args.steps * 1e3], device=device, dtype=torch.float64))
        per_rank = [float(u.item()) for u in g]

    kern_ms, kern_cycles, sclk_mhz = ps.kernel_times()
    n_kernel_samples = ps.n_samples
    ps.release_events()
    # every rank's granted shader clock and attention-kernel time (eight GPUs under load are not granted one clock)
    per_rank_sclk, per_rank_kern = [sclk_mhz], [kern_ms]
    if dist is not None:
        g = [torch.zeros(2, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, torch.tensor([sclk_mhz or 0.0, kern_ms or 0.0], device=device, dtype=torch.float64))
        per_rank_sclk = [float(u[0].item()) or None for u in g]
        per_rank_kern = [float(u[1].item()) or None for u in g]

    parity = None
    if rank == 0 and not args.no_parity:
        scenes = sorted({0, B // 3, (2 * B) // 3, B - 1})
        parity = parity_check(fwd.out, (qm, km, vm, ex), ak, cross, scenes)

    # forward + backward of the operator (gta_attn_bwd: q-side pre-pass, dQ, dK/dV), reported beside the headline
    fwd_bwd_ms, extra_errors = None, {}
    if args.train_steps > 0:
        try:
            qg, kg, vg = (t.detach().clone().requires_grad_() for t in (q, k, v))
            tcg = tc.detach().clone().requires_grad_() if tc is not None else None
            e2 = dict(exd)
            gta_amd.pre_compute_reps_encoder(ak, e2)
            if cross:
                gta_amd.pre_compute_reps_decoder(ak, e2)
            packed = gta_amd.pack_reps(e2, f_dims)
            w = torch.randn_like(q)

            def train_step():
                out = gta_amd.gta_attention(qg, kg, vg, f_dims, packed, so3_degree=e2.get("gta_so3_degree", 0), trans_coeff=tcg,
                                            precise=args.precise, **({"kv_mode": args.kv_mode} if args.kv_mode.startswith("prepass_bwd") else {}))
                out.backward(w)
                qg.grad = kg.grad = vg.grad = None
            precondition(train_step, min(0.6, args.precondition_s), 5)
            for _ in range(3):
                train_step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.train_steps):
                train_step()
            e1.record()
            torch.cuda.synchronize()
            fwd_bwd_ms = e0.elapsed_time(e1) / args.train_steps
        except Exception as e:  # noqa: BLE001   (an extra leg: its failure is reported, the headline line is still printed)
            extra_errors["fwd_bwd"] = f"{type(e).__name__}: {str(e)[:300]}"
    block_layer = None
    if args.block_steps > 0 and rank == 0 and args.workload == "ms-enc" and not args.dry_run:
        try:
            block_layer = block_layer_leg(B, args.block_steps, device, min(0.6, args.precondition_s))
        except Exception as e:  # noqa: BLE001
            extra_errors["block_layer"] = f"{type(e).__name__}: {str(e)[:300]}"
    workloads = None
    wl_names = ([w for w in ("ms-dec", "cl-enc", "cl-dec", "dit") if w != args.workload]
                if (args.workloads == "auto" and args.workload == "ms-enc" and args.dtype == "bf16" and not fused and args.kv_mode == "prepass" and world <= 1)
                else [] if args.workloads in ("auto", "none", "") else [w for w in args.workloads.split(",") if w])
    if wl_names and rank == 0:
        # the other BASELINE workloads (configs 2, 3's decoder, 5) on the same box, right behind the headline: what only the builder's
        # `workloads.jsonl` held before.  The headline's tensors are released first (ms-dec alone holds 0.6 GB).
        workloads = {}
        for w in wl_names:
            try:
                workloads[w] = workload_leg(w, args.dtype, device, L, seed=1234 + rank, precondition_s=args.precondition_s)
            except Exception as e:  # noqa: BLE001   (an extra leg: reported, the headline line is still printed)
                workloads[w] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if args.workloads == "auto":
            # the fp32-faithful mode at the CLEVR-TR encoder shape (runs/clevrtr/GTA/gta/config.yaml:55 mixed_prec: False): fp32 inputs
            try:
                workloads["cl-enc-f32-faithful"] = workload_leg("cl-enc", "f32", device, L, seed=1234 + rank, precise=True,
                                                                precondition_s=args.precondition_s)
            except Exception as e:  # noqa: BLE001
                workloads["cl-enc-f32-faithful"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            torch.cuda.empty_cache()
            try:
                workloads["render-cl"] = render_leg(device, precondition_s=min(0.6, args.precondition_s))
            except Exception as e:  # noqa: BLE001
                workloads["render-cl"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            torch.cuda.empty_cache()
    srt_train = None
    if args.model_train_steps > 0:
        # whole-model optimizer step (conv stem, 5 + 2 Transformer blocks on the HIP attention path,
        # render MLP, MSE loss, AdamW; bf16 autocast; DDP's bucketed gradient all-reduce over RCCL when world > 1)
        # (a failure of this extra leg must not take the headline line with it: it is reported, the line is still printed)
        try:
            from gta_amd import srt
            torch.manual_seed(1234 + rank)
            model = srt.TransformingSRT(srt.msn_gta_so3_cfg()).to(device)
            batch = srt.synthetic_batch(args.model_batch, device=device, seed=99 + rank)
            srt_train = model_train_leg(args, dist, world, rank, local_rank, device, model,
                                        lambda m: srt.compute_loss(m, batch, mixed_prec=(args.dtype == "bf16"))[0].mean())
            srt_train.update({"enc_mtokens_s": srt_train["scenes_per_s"] * 1280 / 1e6,
                              "config": f"MSN gta_so3 TransformingSRT (encoder 5 blocks d=768, decoder 2 blocks, 5 input + 5 target "
                                        f"views, 1280 scene tokens, 2560 query rays), {args.model_batch} scenes/GPU, "
                                        f"AdamW, {args.dtype} autocast, dp{max(world, 1)}"})
            del model, batch
        except Exception as e:  # noqa: BLE001
            srt_train = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    flops = 4.0 * B * H * Tq * Tk * dh                               # QK^T + PV, 2 flop/MAC
    alg_bytes = (2 * Tq + 2 * Tk) * H * dh * q.element_size() * B    # read Q,K,V once, write O once
    achieved = flops / (kern_ms * 1e-3) / 1e12 if kern_ms else None
    traffic, mfma_busy_sq, traffic_src = latest_traffic(args.workload, B, args.dtype, kname) if not fused else (None, None, None)

    if rank == 0:
        n = max(world, 1)
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "GTA-attn Mtokens/s (V=5,H=W=128,d=768)" if args.workload == "ms-enc"
                      else f"GTA-attn Mtokens/s ({args.workload})",
            "value": n * B * Tq * args.steps / elapsed / 1e6,
            "unit": "Mtokens/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: GTA attention forward (rep build + K/V rep pre-pass + attention kernel), "
                                   f"B={B}/GPU, H={H}, Tq={Tq}, Tk={Tk}, dh={dh}, f_dims={f_dims}, "
                                   f"views q/k={Nq}/{Nk}" + (", fp32-faithful products (GTA_FLAG_FP32_PRODUCTS)" if args.precise else ""),
                       "global_batch": n * B, "parallelism": f"dp{n}"},
            "protocol": ("sustained: K steps behind W warmup steps behind ~%.1f s of the same step untimed (r05 on); `cold_start` = rounds 1-4's protocol, same run"
                         % args.precondition_s) if args.precondition_s > 0 else "cold: W warmup + K steps from whatever state the GPU was in (rounds 1-4's protocol)",
            "preconditioning": precond, "cold_start": cold_start,
            "host_ms_per_step": t_host / args.steps * 1e3, "per_rank_ms_per_step": per_rank, "extra_leg_errors": extra_errors or None,
            "per_rank_sclk_mhz": per_rank_sclk, "per_rank_kernel_ms": per_rank_kern,
            "dist": __import__("gta_amd.ddp", fromlist=["backend_info"]).backend_info(),
        }
        if achieved is not None:
            line["roofline"] = {"bound": "mfma", "kernel": kname, "rows_per_item": rows_it, "achieved": achieved,
                                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                                "kernel_ms": kern_ms, "kernel_samples": n_kernel_samples, "algorithmic_flops": flops, "algorithmic_bytes": alg_bytes,
                                "hbm_frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "step_frac": flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                # box-independent view of the same launches (per-item s_memtime / s_memrealtime stamps): a move in
                                # kernel_cycles is the code, a move in sclk_mhz is the clock this box granted
                                "kernel_cycles": kern_cycles, "sclk_mhz": sclk_mhz,
                                "mfma_busy": (flops / MFMA_FLOP_PER_CYCLE / kern_cycles) if kern_cycles else None,
                                "mfma_busy_note": "matrix-pipe cycles the launch's MFMAs need / kernel_cycles (derived); "
                                                  "mfma_busy_sq = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES-per-SIMD) of the committed counter pass",
                                "mfma_busy_sq": mfma_busy_sq,
                                "frac_at_granted_clock": (achieved / (PEAK_BF16_TFLOPS * sclk_mhz / PEAK_CLOCK_MHZ)) if sclk_mhz else None}
        if parity is not None:
            line["parity"] = parity
        if fwd_bwd_ms is not None:
            line["fwd_bwd"] = {"ms_per_step": fwd_bwd_ms, "mtokens_s_per_gpu": B * Tq / (fwd_bwd_ms * 1e-3) / 1e6,
                               "tflops": 3.5 * flops / (fwd_bwd_ms * 1e-3) / 1e12,
                               "note": "operator forward + backward (5 GEMMs + 2 recomputed = 3.5x forward flops), "
                                       "not part of `value`"}
        if block_layer is not None:
            line["block_layer"] = block_layer
        if workloads is not None:
            line["workloads"] = workloads
        if srt_train is not None:
            line["srt_train"] = srt_train
        if not args.no_cpu_baseline and n == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload, 99)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
