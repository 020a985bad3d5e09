"""Audit of the persistent attention kernel's register allocation (gta_fwd2_kernel).

The kernel is an item loop around a tile loop that needs the whole 256-register budget of a wave.  Three things make
hipcc spill there (gta_fwd2.hip says where each is handled): kernel arguments kept live across the loop nest, lane-derived
address arithmetic hoisted out of the item loop, and loop-body arrays that are assigned only conditionally (they become
loop-carried).  A spill of a freshly loaded value sits behind an s_waitcnt vmcnt(0) and serialises the prologue's loads,
so the shipped layouts must stay at ZERO scratch accesses.  This script compiles gta_fwd2.hip to assembly with the
Makefile's flags and reports, per instantiation, registers, scratch accesses and SGPR spill traffic."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (dhp, esz, layout) of the shipped configs: MSN gta_so3 (96, MS = 1), CLEVR-TR gta (64, CL = 2), pure so2 (64, SO2 = 3)
SHIPPED = {(96, 2, 1), (96, 4, 1), (64, 2, 2), (64, 4, 2), (64, 2, 3), (64, 4, 3)}


def audit():
    src = os.path.join(ROOT, "gta_amd", "csrc", "gta_fwd2.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fwd2.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-slp-vectorize", "-S",
                        "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    report, problems = [], []
    for m in re.finditer(r"^(_ZN\w*gta_fwd2_kernelILi(\d+)ELi(\d+)ELi(\d+)E\w+):", text, re.M):
        name, key = m.group(1), (int(m.group(2)), int(m.group(3)), int(m.group(4)))
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        meta = text[text.index(".amdhsa_kernel " + name):][:4000]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        row = {"instance": key, "vgpr": vgpr, "scratch": body.count("scratch_"), "sgpr_spill_writes": body.count("v_writelane"),
               "sgpr_spill_reads": body.count("v_readlane")}
        report.append(row)
        if key in SHIPPED and row["scratch"]:
            problems.append(f"gta_fwd2_kernel<{key}> has {row['scratch']} scratch accesses")
        if key[0] <= 64 and vgpr > 168:
            problems.append(f"gta_fwd2_kernel<{key}> needs {vgpr} VGPRs: the dh <= 64 instances must allow three waves per SIMD")
    return report, problems


if __name__ == "__main__":
    rep, prob = audit()
    for r in rep:
        print(r)
    print("problems:", prob or "none")
    sys.exit(1 if prob else 0)
