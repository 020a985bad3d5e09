#!/usr/bin/env python3
"""Where the slot-time of an attention launch goes (developer tool, r05): the 32-row kernel stamps every work item with its start / end (shader
cycles and the 100-MHz clock) and with the CU it ran on (HW_ID | XCC_ID << 32); this rebuilds each CU's occupancy over the launch.

    python tools/wg_timeline.py cl-dec,cl-enc [persist|-] [workgroups per CU = 3]

Per workload: span of the launch, item durations (percentiles; first-round items against later ones), resident workgroups over time (16 bins),
per-CU number of items / busy time / last end, and the share of slot-time that is inside no item (dispatch gaps, start-up skew, the tail)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gta_amd import native


def pct(t, qs=(0.05, 0.5, 0.95, 1.0)):
    t = t.sort().values
    return [float(t[min(len(t) - 1, int(q * (len(t) - 1) + 0.5))]) for q in qs]


def main():
    wls = (sys.argv[1] if len(sys.argv) > 1 else "cl-dec").split(",")
    persist = len(sys.argv) > 2 and sys.argv[2] == "persist"
    dev = torch.device("cuda", 0)
    L = native.lib()
    for wl in wls:
        rows32 = wl.endswith(":rows32")
        name = wl.split(":")[0]
        fl = (native.FLAG_ROWS32 if rows32 else 0) | (native.FLAG_PERSIST if persist else 0)
        ps = bench.PlannedStep(name, bench.WORKLOADS[name][8], "bf16", dev, L, seed=1, steps=1, kernel_samples=1, flags=fl, time_kernel=False)
        vq, vk, cq, ck = ps.build_reps()
        ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc)
        torch.cuda.synchronize()
        n_it, rows_it = ctypes.c_int32(0), ctypes.c_int32(0)
        kname = (L.gta_debug_attention_kernel(ctypes.byref(ps.fwd.desc), ctypes.byref(n_it), ctypes.byref(rows_it)) or b"").decode()
        prof = torch.zeros(max(n_it.value, 1), 8, dtype=torch.int64, device=dev)

        def run():
            return ps.fwd(ps.q, ps.k, ps.v, vq, vk, cq, ck, ps.tc, flags_extra=native.FLAG_KV_READY)
        bench.precondition(run, 1.0)
        L.gta_debug_profile_next_attention_kernel(ctypes.c_void_p(prof.data_ptr()), prof.shape[0])
        run()
        torch.cuda.synchronize()
        P = prof.cpu()
        ok = P[:, 6] > P[:, 5]
        P = P[ok]
        cyc, mhz = bench.kernel_clock(prof)
        t0 = int(P[:, 5].min())
        st, en = (P[:, 5] - t0).double() / 100.0, (P[:, 6] - t0).double() / 100.0          # microseconds
        span = float(en.max())
        dur = en - st
        hw = P[:, 1]
        xcc = (hw >> 32) & 0xf
        lo = hw & 0xffffffff
        cu, sh, se = (lo >> 8) & 0xf, (lo >> 12) & 0x1, (lo >> 13) & 0x7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        print(f"== {wl}{' persist' if persist else ''}: {kname}, {len(P)} items of {rows_it.value} rows; span {span:.1f} us = {cyc / 1e3:.1f}k cycles at {mhz:.0f} MHz")
        print(f"   item us p5/p50/p95/max: {['%.1f' % x for x in pct(dur)]}; mean {float(dur.mean()):.2f}; sum/span = {float(dur.sum()) / span:.0f} resident on average")
        first = st < 0.25 * float(dur.median())
        print(f"   first-round items ({int(first.sum())}): mean {float(dur[first].mean()):.2f} us; later: {float(dur[~first].mean()):.2f} us; "
              f"start skew of the first round p95 {pct(st[first])[2]:.2f} us")
        bins = 16
        occ = []
        for b in range(bins):
            a, e = span * b / bins, span * (b + 1) / bins
            occ.append(float(((en.clamp(max=e) - st.clamp(min=a)).clamp(min=0)).sum()) / (e - a))
        print("   resident workgroups over time:", " ".join(f"{o:.0f}" for o in occ))
        ids = cuid.unique()
        per = []
        for c in ids.tolist():
            m = cuid == c
            per.append((int(m.sum()), float(dur[m].sum()), float(en[m].max()), float(st[m].min())))
        T = torch.tensor(per)
        print(f"   CUs seen {len(ids)}: items per CU min/med/max {int(T[:, 0].min())}/{int(T[:, 0].median())}/{int(T[:, 0].max())}; "
              f"busy slot-us per CU p5/p50/p95/max {['%.0f' % x for x in pct(T[:, 1])]}; last end per CU p5/p50/p95/max {['%.1f' % x for x in pct(T[:, 2])]}")
        # per-CU gaps: sort a CU's items by start, pack them greedily on its slots, sum the idle time between an end and the next start
        slots = int(sys.argv[3]) if len(sys.argv) > 3 else 3          # resident workgroups per CU (3 at dh <= 64, 2 at dh = 96)
        gap_tot, n_gap = 0.0, 0
        for c in ids.tolist():
            m = (cuid == c).nonzero().flatten()
            order = m[st[m].argsort()]
            ends = []
            for i in order.tolist():
                s_i = float(st[i])
                done = [e for e in ends if e <= s_i + 1e-9]
                if done and len(ends) >= slots:
                    e_best = max(done)
                    gap_tot += s_i - e_best
                    n_gap += 1
                    ends.remove(e_best)
                ends.append(float(en[i]))
        print(f"   end -> next start on the same CU: {n_gap} hand-overs, mean gap {gap_tot / max(n_gap, 1):.2f} us ({gap_tot / max(n_gap, 1) * mhz / 1e3:.1f}k cycles); "
              f"idle share of slot-time: {1.0 - float(dur.sum()) / (span * len(ids) * slots):.3f}")
        wid = lo & 0xf
        print("   wave slot of wave 0 (HW_ID[3:0]) -> items / mean us:", " ".join(f"{w}:{int((wid == w).sum())}/{float(dur[wid == w].mean()):.1f}" for w in wid.unique().tolist()))
        xs = []
        for x in range(8):
            m = xcc == x
            if bool(m.any()):
                xs.append(f"{x}:{int(m.sum())}/{float(dur[m].mean()):.1f}/{float(en[m].max()):.0f}")
        print("   per XCD items/mean us/last end:", " ".join(xs))
        del ps
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
