"""Audit of the asm-owned accumulator file of the pipelined flash kernel (gta_fwd3_kernel).

The kernel addresses a[16:255] by literal number inside asm statements; hipcc does not know and may park
values of its own in AGPRs.  This script compiles gta_fwd3.hip to assembly with the Makefile's flags and
checks, for every instantiation: no scratch, no VGPR spills, and every compiler-generated AGPR access
(outside ;;#ASMSTART/;;#ASMEND) stays below a16.  Exit code 0 = clean.  (cdna_hip_programming.md 5.7 item 4)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESERVED = 16


def audit(extra_flags=()):
    src = os.path.join(ROOT, "gta_amd", "csrc", "gta_fwd3.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fwd3.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
               *extra_flags, src, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    problems, report = [], []
    for name in re.findall(r"^(_ZN\w*gta_fwd3_kernel\w+):", text, re.M):
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        inasm, worst = False, -1
        for line in body.split("\n"):
            if "ASMSTART" in line:
                inasm = True
            elif "ASMEND" in line:
                inasm = False
            elif not inasm:
                code = line.split(";")[0]
                for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", code):
                    worst = max(worst, int(m.group(2) or m.group(1)))
        meta = text[text.index(".name:           " + name) - 1500:text.index(".name:           " + name) + 600]
        spills = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", meta[1500:]).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", meta[1500:]).group(1))
        report.append((name, worst, spills, scratch))
        if worst >= RESERVED or spills or scratch:
            problems.append((name, worst, spills, scratch))
    return report, problems


if __name__ == "__main__":
    report, problems = audit()
    for name, worst, spills, scratch in report:
        print(f"{name[-40:]:40s} highest compiler-touched AGPR: a{worst}  vgpr_spills={spills} scratch={scratch}")
    if problems:
        print("AUDIT FAILED:", problems)
        sys.exit(1)
    print(f"ok: {len(report)} instantiations, compiler stays below a{RESERVED}")
