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
    for m in re.finditer(r"^(_ZN\w*gta_fwd2_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E\w+):", text, re.M):
        x3 = m.group(5) == "1"          # the fp32-faithful instances (split-bf16 operands): two workgroups per CU, 256 registers
        name, key = m.group(1), (int(m.group(2)), int(m.group(3)), int(m.group(4))) + (("x3",) if x3 else ())
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        meta = text[text.index(".amdhsa_kernel " + name):][:4000]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        row = {"instance": key, "vgpr": vgpr, "scratch": body.count("scratch_"), "sgpr_spill_writes": body.count("v_writelane"),
               "sgpr_spill_reads": body.count("v_readlane")}
        report.append(row)
        if (key in SHIPPED or (x3 and key[:3] in SHIPPED)) and row["scratch"]:
            problems.append(f"perf: gta_fwd2_kernel<{key}> has {row['scratch']} scratch accesses")
        if key[0] <= 64 and vgpr > (256 if x3 else 168):
            problems.append(f"perf: gta_fwd2_kernel<{key}> needs {vgpr} VGPRs: the dh <= 64 instances must allow three waves per SIMD")
    # the dh = 64 bf16 instance (gta_fwd_cl.hip, r06): three workgroups per CU, no scratch anywhere
    text = _asm("gta_fwd_cl.hip", ("-fno-slp-vectorize",))
    for name, (layout,), body, vgpr in _kernels(text, r"gta_fwdc_kernelILi(\d+)E"):
        row = {"kernel": f"gta_fwdc_kernel<{layout}>", "vgpr": vgpr, "scratch": body.count("scratch_"), "sgpr_spill_writes": body.count("v_writelane")}
        report.append(row)
        if row["scratch"] and int(layout) == 2:
            problems.append(f"perf: {row['kernel']}: {row['scratch']} scratch accesses")
        if vgpr > 168:
            problems.append(f"perf: {row['kernel']}: {vgpr} VGPRs (three waves per SIMD need <= 168)")
    return report, problems


def _asm(src_name, extra=()):
    src = os.path.join(ROOT, "gta_amd", "csrc", src_name)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", *extra, "-S",
                        "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def _kernels(text, pattern):
    for m in re.finditer(r"^(_ZN\w*" + pattern + r"\w*):", text, re.M):
        name = m.group(1)
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        meta = text[text.index(".amdhsa_kernel " + name):][:4000]
        yield name, m.groups()[1:], body, int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))


def audit_others():
    """The kernels beside the attention forward that were tuned by instruction counts and occupancy (DESIGN.md sections 4.1,
    4.3, 8): the K/V pre-pass must stay scratch-free and within the registers of five workgroups per CU at the shipped head
    sizes (it is at 89 VGPRs at dh = 96: the LDS would allow six), the backward kernels scratch-free (dK/dV at dh = 128 runs one workgroup per CU for that), and the weight-gradient kernel's steady-state loop (the loops with
    one step's 32 MFMAs) free of scratch accesses."""
    report, problems = [], []
    text = _asm("gta_prep.hip", ("-fno-slp-vectorize",))
    for name, (dhp, esz), body, vgpr in _kernels(text, r"gta_kv_prep_kernelILi(\d+)ELi(\d+)E"):
        row = {"kernel": f"gta_kv_prep_kernel<{dhp},{esz}>", "vgpr": vgpr, "scratch": body.count("scratch_")}
        report.append(row)
        if row["scratch"]:
            problems.append(f"perf: {row['kernel']}: {row['scratch']} scratch accesses")
        if int(dhp) <= 96 and int(esz) == 2 and vgpr > 96:
            problems.append(f"perf: {row['kernel']}: {vgpr} VGPRs (five 4-wave workgroups per CU need <= 96; six would need <= 80)")
    text = _asm("gta_bwd.hip", ("-fno-slp-vectorize",))
    for kern in ("gta_bwd_prep_kernel", "gta_bwd_dq_kernel", "gta_bwd_dkv_kernel"):
        for name, (dhp, esz), body, vgpr in _kernels(text, kern + r"ILi(\d+)ELi(\d+)E"):
            row = {"kernel": f"{kern}<{dhp},{esz}>", "vgpr": vgpr, "scratch": body.count("scratch_")}
            report.append(row)
            if row["scratch"]:
                problems.append(f"perf: {row['kernel']}: {row['scratch']} scratch accesses")
    text = _asm("gta_wgrad.hip")
    for name, _, body, vgpr in _kernels(text, r"wgrad_kernelILb(\d)E"):
        lines = body.split("\n")
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        steady = 0
        for i, l in enumerate(lines):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loop = lines[labels[m.group(1)]:i]
                if sum("v_mfma" in x for x in loop) == 32 and len(loop) < 400:
                    steady += 1
                    if any("scratch_" in x for x in loop):
                        problems.append(f"perf: {name}: scratch access inside a steady-state loop")
        report.append({"kernel": name, "vgpr": vgpr, "steady_loops": steady})
        if not steady:
            problems.append(f"perf: {name}: no loop with one step's 32 MFMAs found")
    return report, problems


def audit_dkv64():
    """gta_bwd_dkv64_kernel (gta_bwd.hip): ONE generated statement (gen_bwd64.py) owns v24..v255 and every accumulator register; it leaves
    dK'^T / dV'^T in a[0:191] (the dQ kernel: a[0:95]) and names them as its OUTPUTS (GTA_BWD64_*_RESULTS, r05), so hipcc reads them out
    itself.  What it may do with the accumulator file: v_accvgpr_read of those result registers, nothing else (no writes, no copies, no
    other register); and at most a handful of scratch accesses (the epilogue holds 192 accumulator values beside its own)."""
    text = _asm("gta_bwd.hip", ("-fno-slp-vectorize",))
    report, problems = [], []
    for m in re.finditer(r"^(_ZN\w*gta_bwd_(?:dkv|dq|dqkv)64_kernel\w+):", text, re.M):
        name = m.group(1)
        want_stmts = 2 if "dqkv64" in name else 1            # (the joint launch holds both bodies)
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        meta = text[text.index(".amdhsa_kernel " + name):][:4000]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        accum = int(re.search(r"\.amdhsa_accum_offset\s+(\d+)", meta).group(1))
        n_results = 288 if "dqkv64" in name else 192 if "dkv64" in name else 96        # (the joint kernel: either body's registers)
        n_results = min(n_results, 192)
        inasm, compiler_acc, scratch, long_stmts, cur_len, result_reads = False, 0, 0, 0, 0, 0
        for line in body.split("\n"):
            if "#ASMSTART" in line:
                inasm, cur_len = True, 0
                continue
            if "#ASMEND" in line:
                inasm = False
                long_stmts += cur_len > 1000
                continue
            if inasm:
                cur_len += 1
                continue
            code = line.split(";")[0]
            n_acc = len(re.findall(r"\ba\[?\d+", code))
            mr = re.match(r"\s*v_accvgpr_read_b32\s+v\d+,\s*a(\d+)\s*$", code)
            if mr and int(mr.group(1)) < n_results:
                result_reads += 1
            else:
                compiler_acc += n_acc
            scratch += "scratch_" in line
        report.append({"kernel": name, "vgpr": vgpr, "accum_offset": accum, "result_reads": result_reads, "compiler_agpr_uses": compiler_acc,
                       "scratch": scratch, "loop_statements": long_stmts})
        if compiler_acc:
            problems.append(f"{name}: hipcc uses accumulator registers beyond reading the statements' results ({compiler_acc} operands)")
        want_reads = {"dqkv64": 288, "dkv64": 192}.get("dqkv64" if "dqkv64" in name else "dkv64" if "dkv64" in name else "", 96)
        if result_reads != want_reads:
            problems.append(f"{name}: {result_reads} reads of result registers (expected {want_reads}: every result once)")
        if long_stmts != want_stmts:
            problems.append(f"{name}: {long_stmts} generated statements (expected {want_stmts})")
        if vgpr != 512 or accum != 256:
            problems.append(f"{name}: register file split {accum} / {vgpr} (expected 256 / 512)")
        if scratch > 12 * want_stmts:
            problems.append(f"perf: {name}: {scratch} scratch accesses")
    return report, problems


def audit_attn64():
    """gta_attn64_kernel (gta_fwd64.hip): the tile loop statement owns v32-v255 and the accumulator file from a28 up by literal register
    number.  Since r05 what crosses its boundary in the accumulator file is in its operand list -- the Q' fragments it takes ("+{a[..]}"), the O
    accumulators it leaves ("={a[..]}"; gen_attn64.py) -- so hipcc moves those values itself and may park others in a[0:27] or anywhere the
    statement does not name; the audit counts that traffic for the record.  Checked: the loop statement is the single long one, the register
    file split, and (nearly) no scratch access (a reload sits behind a vmcnt(0))."""
    text = _asm("gta_fwd64.hip", ("-fno-slp-vectorize",))
    report, problems = [], []
    for m in re.finditer(r"^(_ZN\w*gta_attn64_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E\w+):", text, re.M):
        name, key = m.group(1), tuple(int(m.group(i)) for i in range(2, 7))       # (dh, element size, layout, variant, coalesced I/O)
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        meta = text[text.index(".amdhsa_kernel " + name):][:4000]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        accum = int(re.search(r"\.amdhsa_accum_offset\s+(\d+)", meta).group(1))
        inasm, compiler_acc, n_stmt_lines, scratch_lines, long_stmts, cur_len = False, 0, 0, [], 0, 0
        for i, line in enumerate(body.split("\n")):
            if "#ASMSTART" in line:
                inasm, cur_len = True, 0
                continue
            if "#ASMEND" in line:
                inasm = False
                long_stmts += cur_len > 1000
                continue
            if inasm:
                cur_len += 1
                continue
            for m2 in re.finditer(r"\ba\[?(\d+)(?::(\d+))?", line.split(";")[0]):
                compiler_acc += int(m2.group(2) or m2.group(1)) >= 28          # (a[0:27] are left to hipcc: gen_attn64.py)
            if "scratch_" in line:
                scratch_lines.append(i)
        row = {"instance": key, "vgpr": vgpr, "accum_offset": accum, "compiler_agpr_uses": compiler_acc, "scratch": len(scratch_lines),
               "loop_statements": long_stmts}
        report.append(row)
        if long_stmts != 1:
            problems.append(f"gta_attn64_kernel<{key}>: {long_stmts} loop statements (expected one)")
        if vgpr != 512 or accum != 256:
            problems.append(f"gta_attn64_kernel<{key}>: register file split {accum} / {vgpr} (expected 256 / 512)")
        if len(scratch_lines) > 16:
            problems.append(f"perf: gta_attn64_kernel<{key}>: {len(scratch_lines)} scratch accesses")
    # gta_attn64_items_kernel: the whole item loop is ONE statement that names v8..v255, every accumulator register and s20..s99; what
    # hipcc keeps across it lives in v0..v7 (SGPR spills go to lanes of those) -- no scratch, nothing of hipcc's in the accumulator file
    for m in re.finditer(r"^(_ZN\w*gta_attn64_items_kernel\w+):", text, re.M):
        name = m.group(1)
        body = text[m.start():text.index(".Lfunc_end", m.start())]
        meta = text[text.index(".amdhsa_kernel " + name):][:4000]
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
        accum = int(re.search(r"\.amdhsa_accum_offset\s+(\d+)", meta).group(1))
        inasm, compiler_acc, long_stmts, cur_len, hi_vgpr = False, 0, 0, 0, 0
        for line in body.split("\n"):
            if "#ASMSTART" in line:
                inasm, cur_len = True, 0
                continue
            if "#ASMEND" in line:
                inasm = False
                long_stmts += cur_len > 1000
                continue
            if inasm:
                cur_len += 1
                continue
            code = line.split(";")[0]
            compiler_acc += len(re.findall(r"\ba\[?\d+", code))
        row = {"instance": "items", "vgpr": vgpr, "accum_offset": accum, "compiler_agpr_uses": compiler_acc, "scratch": body.count("scratch_"),
               "loop_statements": long_stmts}
        report.append(row)
        if compiler_acc or row["scratch"] or long_stmts != 1 or vgpr != 512 or accum != 256:
            problems.append(f"gta_attn64_items_kernel: {row}")
    return report, problems


if __name__ == "__main__":
    rep, prob = audit()
    rep2, prob2 = audit_others()
    rep3, prob3 = audit_attn64()
    rep4, prob4 = audit_dkv64()
    for r in rep + rep2 + rep3 + rep4:
        print(r)
    #  the build fails on the CORRECTNESS criteria only -- what hipcc does to the accumulator file around the generated statements, the
    # register-file split, statement and result-read counts; scratch and register counts are performance criteria sitting close to today's
    # compiler output: reported as warnings ("perf:"), with headroom
    allp = prob + prob2 + prob3 + prob4
    hard = [x for x in allp if not x.startswith("perf:")]
    print("warnings:", [x for x in allp if x.startswith("perf:")] or "none")
    print("problems:", hard or "none")
    sys.exit(1 if hard else 0)
