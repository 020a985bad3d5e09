#!/usr/bin/env python3
"""The K/V pre-pass launch alone at a bench workload, sustained clock (developer tool): microseconds per launch by stream events over
back-to-back GTA_FLAG_PREP_ONLY calls behind a second of the same.  GTA_HIP_LIB selects the library (tools/ablate_prep.sh: the ablation builds).

    python tools/time_prep.py [ms-enc] [label]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gta_amd import native


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "ms-enc"
    label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("GTA_HIP_LIB", "libgta_hip.so"))
    dev = torch.device("cuda", 0)
    H, Nq, Pq, Nk, Pk, f_dims, so2, so3, B = bench.WORKLOADS[wl]
    ps = bench.PlannedStep(wl, B, "bf16", dev, native.lib(), seed=1, steps=1, kernel_samples=1, time_kernel=False)
    vq, vk, cq, ck = ps.build_reps()
    q = ps.q

    def run():
        ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc, flags_extra=native.FLAG_PREP_ONLY)
    bench.precondition(run, 1.0)
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 200 * 1e3)
    mb = 2 * 2 * B * H * Nk * Pk * q.shape[-1] * 2 / 1e6
    print(f"{label:28s} {wl}: " + " ".join(f"{u:6.2f}" for u in res) + f" us per launch; K, V in + images out = {mb:.0f} MB -> {mb / min(res):.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
