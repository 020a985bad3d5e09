"""Gaps between the kernels of a bench step, from a rocprofv3 --kernel-trace CSV.

usage: python tools/step_gaps.py <dir with *_kernel_trace.csv> [anchor kernel substring]
Prints, over the steady-state steps, each kernel's mean duration and the mean idle gap in front of it, and
the step period (start of an anchor kernel to the next)."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(gta_\w+|build_\w+)", name)
    return m.group(1) if m else name[:40]


def main(src, anchor="build_reps"):
    import os
    files = sorted(glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)     # newest last
    rows = []
    for r in csv.DictReader(open(files[-1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < 6:
        print("too few anchor kernels:", len(idx)); return
    idx = idx[len(idx) // 3:]                       # steady state: drop the first third (warm-up)
    periods = [rows[b][0] - rows[a][0] for a, b in zip(idx, idx[1:])]
    med = sorted(periods)[len(periods) // 2]
    steps = [(a, b) for a, b in zip(idx, idx[1:]) if rows[b][0] - rows[a][0] < 1.5 * med]
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in steps:
        for i in range(a, b):
            dur[rows[i][2]].append(rows[i][1] - rows[i][0])
            gap[rows[i][2]].append(rows[i][0] - rows[i - 1][1])
    print(f"{len(steps)} steps; step period median {med / 1e3:.1f} us")
    tot_d = tot_g = 0
    for k in dur:
        d, g = sum(dur[k]) / len(steps), sum(gap[k]) / len(steps)
        tot_d += d; tot_g += g
        print(f"  {k:28s} {len(dur[k]) / len(steps):4.1f} launches/step  kernel {d / 1e3:7.1f} us   idle in front {g / 1e3:6.1f} us")
    print(f"  sum of kernels {tot_d / 1e3:.1f} us + gaps {tot_g / 1e3:.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
