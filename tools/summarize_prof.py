"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into small committed files under profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof_v4 profiles/r01/v4
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section): `fetch_bytes_corrected` doubles it.
WRITE_SIZE is taken as is (uncalibrated, see the same section).
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(gta_\w+|build_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def latest(files):
    """gpurun merges every call's outputs into the same local directory: keep the newest run (by modification time; the
    numeric prefix is a process id and says nothing about order) of each directory only."""
    import os
    by_dir = collections.defaultdict(list)
    for f in files:
        by_dir[f.rsplit("/", 1)[0]].append((os.path.getmtime(f), f))
    return [max(v)[1] for v in by_dir.values()]


def main(src, dst):
    out = {}
    for f in latest(glob.glob(f"{src}/stats/**/*_kernel_stats.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                out.setdefault(k, {})["calls"] = int(r["Calls"])
                out[k]["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
                out[k]["min_us"] = round(float(r["MinNs"]) / 1e3, 2)
                out[k]["pct_of_gpu_time"] = float(r["Percentage"])
    for f in latest(glob.glob(f"{src}/pmc_*/**/*_counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            for c, v in d.items():
                out.setdefault(k, {})[c] = round(sum(v) / len(v), 1)
    for k, d in out.items():
        if "FETCH_SIZE" in d:
            d["fetch_bytes_corrected"] = int(d["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in d:
            d["write_bytes"] = int(d["WRITE_SIZE"] * 1024)
    json.dump(out, open(dst + "_kernels.json", "w"), indent=1, sort_keys=True)
    with open(dst + "_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,avg_us,min_us,pct_of_gpu_time,fetch_bytes_corrected,write_bytes\n")
        for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("pct_of_gpu_time", 0)):
            f.write(f"\"{k}\",{d.get('calls','')},{d.get('avg_us','')},{d.get('min_us','')},{d.get('pct_of_gpu_time','')},"
                    f"{d.get('fetch_bytes_corrected','')},{d.get('write_bytes','')}\n")
    print(open(dst + "_kernel_stats.csv").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
