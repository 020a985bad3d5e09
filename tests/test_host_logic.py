"""CPU: host-side logic -- the C ABI library loads and exports every declared symbol, descriptor
validation (no GPU needed), the LDS swizzle is conflict-free, reps packing, module surface."""
import ctypes
import re

import numpy as np
import pytest
import torch

import gta_amd
from gta_amd import native
from tests import _golden as G


def test_library_exports_every_declared_symbol():
    lib = native.lib()
    hdr = open(native.LIB_PATH.replace("gta_amd/csrc/libgta_hip.so", "include/gta_hip.h")).read()
    declared = set(re.findall(r"\b(gta_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(native.ABI_SYMBOLS), declared ^ set(native.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.gta_abi_version() == native.GTA_ABI_VERSION
    assert lib.gta_sizeof_attn_desc() == ctypes.sizeof(native.GtaAttnDesc)


def test_block_library_loads_and_exports_declared_symbols():
    """include/gta_block.h <-> libgta_block.so <-> gta_amd/native_block.py (no compute without a GPU)."""
    from gta_amd import native_block as nb
    lib = nb.lib()
    hdr = open(nb.LIB_PATH.replace("gta_amd/csrc/libgta_block.so", "include/gta_block.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|void|const char\*)\s+(gta_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert declared == set(nb.ABI_SYMBOLS), declared ^ set(nb.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.gta_block_abi_version() == nb.BLOCK_ABI_VERSION
    assert lib.gta_sizeof_gemm_desc() == ctypes.sizeof(nb.GtaGemmDesc)
    assert lib.gta_gemm_workspace_bytes() == 32 << 20
    assert lib.gta_ln_bwd_workspace_bytes(40960, 768) == 1024 * 2 * 768 * 4
    # argument checks run before anything touches a device
    assert lib.gta_ln_fwd(None, 0, None, None, 1e-5, 4, 8, None, 0, None, None, None) == -1
    d = nb.GtaGemmDesc()
    assert lib.gta_gemm(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == -1
    with pytest.raises(native.GtaError):
        nb.ln_fwd(torch.zeros(4, 16), torch.ones(16), torch.zeros(16), 1e-5, torch.bfloat16)     # CPU tensors: no fallback


def test_wgrad_plan_without_gpu():
    """gta_wgrad's shape rule and split-M plan are host code: the workspace covers S partial tiles (+ bias partials), every
    split has at least eight 32-token steps, and the splits cover all tokens."""
    from gta_amd import native_block as nb
    lib = nb.lib()
    assert lib.gta_wgrad_supported(40960, 2304, 768) == 1 and lib.gta_wgrad_supported(64, 256, 256) == 1
    for bad in ((40961, 768, 768), (40960, 700, 768), (40960, 768, 100), (0, 256, 256), (32, 256, 128)):
        assert lib.gta_wgrad_supported(*bad) == 0 and lib.gta_wgrad_workspace_bytes(*bad) == 0, bad
    for m, n, k in ((40960, 2304, 768), (40960, 768, 768), (4160, 768, 256), (64, 256, 512), (256, 256, 256)):
        nbytes = lib.gta_wgrad_workspace_bytes(m, n, k)
        per_split = (n * k + n) * 4
        assert nbytes > 0 and nbytes % per_split == 0
        S, steps_total, tiles = nbytes // per_split, m // 32, (n // 256) * (k // 256)
        assert 1 <= S <= max(1, steps_total // 8) and S * tiles <= 256 + tiles      # about one workgroup per CU
        per = -(-steps_total // S)
        assert per * (S - 1) < steps_total <= per * S                               # every split has work, all tokens covered
    # argument checks run before anything touches a device
    assert lib.gta_wgrad(None, 0, None, 0, 64, 256, 256, None, None, None, 0, None) == -1


def test_fused_blocks_do_not_engage_off_gpu():
    """On CPU tensors the Transformer keeps the module-by-module path (and the attention operator then refuses)."""
    from gta_amd import fused
    assert fused.compute_dtype(torch.zeros(2, 3, 8)) is None
    assert fused.norm_ok(torch.nn.LayerNorm(48)) and fused.norm_ok(torch.nn.LayerNorm(180)) and fused.norm_ok(torch.nn.LayerNorm(20))
    assert not fused.norm_ok(torch.nn.LayerNorm(18)) and not fused.norm_ok(torch.nn.LayerNorm(2052))     # rows of 4k elements, <= 2048
    assert not fused.norm_ok(torch.nn.LayerNorm(48, elementwise_affine=False))


def _desc(dh=96, f=None, L=2, **kw):
    f = f or {"se3": 48, "so3": 24, "so2": 24}
    q = torch.empty(2, 8, 1280, dh)
    d = native.make_desc(q, q, q, q, f, L, 5, 5, dh ** -0.5, native.FLAG_V_TRANSFORM)
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_descriptor_validation_without_gpu():
    ok = native.attn_fwd_supported
    assert ok(_desc()) == 0
    assert ok(_desc(64, {"se3": 32, "so2": 32}, 0)) == 0
    assert ok(_desc(64, {"so2": 64}, 0)) == 0
    assert ok(_desc(64, {"triv": 8, "se3": 24, "so2": 32}, 0)) == 0
    assert ok(_desc(96, {"se3": 48, "so3": 24, "so2": 20}, 2)) == -2          # slabs do not sum to dh
    assert ok(_desc(64, {"se3": 30, "so2": 34}, 0)) == -2                     # se3 not a multiple of 4
    assert ok(_desc(64, {"so2": 56, "t2": 8}, 0)) == -2                       # t2 not a multiple of 3
    assert ok(_desc(96, {"se3": 48, "so3": 24, "so2": 24}, 1)) == -3          # valid (8 x [3]) but not fused
    assert ok(_desc(96, {"se3": 48, "so3": 28, "so2": 20}, 2)) == -2          # so3 not r*(3+5)
    assert ok(_desc(Tq=1281)) == -1                                           # views must split evenly
    d = _desc()
    d.flags |= native.FLAG_EUCLID
    assert ok(d) == -3
    d = _desc(14, {"se3": 6, "so2": 8}, 0)          # the euclid fixture's layout: 3-vectors
    d.flags |= native.FLAG_EUCLID
    assert ok(d) == -3
    assert b"euclid" in native.lib().gta_strerror(-3)
    assert native.launch_info(_desc()) == {"lds_bytes": 86784, "workgroups": 2 * 8 * 10, "threads": 256}  # fp32
    assert native.launch_info(_desc(dtype=native.DTYPE_BF16))["lds_bytes"] == 64256      # bf16: 2 WGs per CU


# ds_read_b128 lane groups on gfx950 (MI355X_MICROARCH.md, LDS table)
B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def swz(units, r, u):
    """Python model of swz<UNITS>() in gta_amd/csrc/gta_common.h."""
    if units == 8:
        rot = 4 * ((r >> 1) & 1) + ((r >> 2) & 3)
    elif units == 16:
        rot = 4 * (r & 3) + ((r >> 2) & 3)
    else:
        tz = 4 if units % 16 == 0 else 3 if units % 8 == 0 else 2 if units % 4 == 0 else 1 if units % 2 == 0 else 0
        rot = (r >> (4 - tz)) & ((1 << tz) - 1)
    return (u + rot) % units


@pytest.mark.parametrize("units", [4, 8, 12, 16, 24, 32])
def test_swizzle_is_a_conflict_free_permutation(units):
    for r in range(64):
        assert sorted(swz(units, r, u) for u in range(units)) == list(range(units))
        assert swz(units, r + 16, 0) == swz(units, r, 0)          # period 16 rows: a slab's offsets are immediates (gta_fwd2.hip)
    # (a) staging reads: lane == row, all lanes read the same logical unit
    # (b) MFMA fragment reads: lanes 0-31 rows 0-31 unit u, lanes 32-63 rows 0-31 unit u+1
    for u in range(units - 1):
        for mode in ("stage", "frag"):
            for grp in B128_GROUPS:
                slots = []
                for lane in grp:
                    row = lane if mode == "stage" else (lane & 31)
                    uu = u if mode == "stage" else u + (lane >> 5)
                    slots.append((row * units + swz(units, row, uu)) % 16)   # 16-B slot in the 256-B bank row
                assert len(set(slots)) == 16, (units, u, mode, slots)


@pytest.mark.parametrize("units", [4, 8, 12, 16])
def test_swizzle_transpose_reads_are_conflict_free(units):
    """ds_read_b64_tr_b16 of a bf16 tile image (the V' / K'^T / Q''^T operands: gta_fwd2.hip voff): lane (lh, g16, p16) reads 8 bytes at key row
    4 lh + (p16 >> 2) + 8 hf (+ 16 per slab), channel unit 4 d + 2 (g16 & 1) + ((p16 & 3) >> 1), half p16 & 1; the instruction runs as two
    32-lane groups of one LDS cycle each (MI355X_MICROARCH.md), bank = (byte address / 4) mod 64 -- no bank twice within a group (r06: the
    8- and 16-unit rotations; before, rows r and r + 2 of an 8-unit image collided: 33 % of the CLEVR-TR kernel's LDS cycles)."""
    for slab in range(4):
        for hf in range(2):
            for d in range(units // 4):
                for grp in range(2):
                    banks = []
                    for lane in range(32 * grp, 32 * grp + 32):
                        lh, g16, p16 = lane >> 5, lane >> 4, lane & 15
                        r = 16 * slab + 4 * lh + (p16 >> 2) + 8 * hf
                        u = 4 * d + 2 * (g16 & 1) + ((p16 & 3) >> 1)
                        a = (r * units + swz(units, r, u)) * 16 + (p16 & 1) * 8
                        banks += [(a // 4) % 64, (a // 4 + 1) % 64]
                    assert len(set(banks)) == 64, (units, slab, hf, d, grp)


def test_pack_reps_from_reference_style_dict():
    d, meta = G.load("op_ms_cross")
    ex = G.extras_of(d, torch.float32)
    packed = gta_amd.pack_reps(ex, meta["f_dims"])
    vq = packed["vrep_q"]
    assert vq.shape == (meta["B"], meta["Nq"], native.VREP_STRIDE)
    assert torch.equal(vq[..., :16].reshape(ex["inv_se3rep_q"].shape), ex["inv_se3rep_q"])
    assert torch.equal(vq[..., 16:32].reshape(ex["se3rep_q"].shape), ex["se3rep_q"])
    assert torch.equal(vq[..., 32:41].reshape(ex["so3rep_q"][0].shape), ex["so3rep_q"][0])
    assert torch.equal(vq[..., 41:66].reshape(ex["so3rep_q"][1].shape), ex["so3rep_q"][1])
    cs = packed["cs_k"]
    assert torch.equal(cs[..., 0], ex["so2rep_k"][..., 0, 0]) and torch.equal(cs[..., 1], ex["so2rep_k"][..., 1, 0])
    assert gta_amd.pack_reps(ex, meta["f_dims"])["vrep_q"] is vq          # cached in the shared dict


@pytest.mark.parametrize("case", G.list_cases("mod_"))
def test_module_state_dict_is_reference_compatible(case):
    d, meta = G.load("mod_" + case)
    ak = {"f_dims": meta["f_dims"], "so2": meta["so2"], "so3": meta["so3"], "max_freq_h": 1, "max_freq_w": 1}
    tr = gta_amd.Transformer(meta["dim"], meta["depth"], meta["H"], meta["dh"], 2 * meta["dim"], 0.0,
                             not meta["cross"], meta["kv_dim"], False, {"method": {"name": "gta", "args": ak}})
    sd = {k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")}
    tr.load_state_dict(sd, strict=True)


def test_no_cpu_fallback():
    q = torch.randn(1, 2, 16, 16)
    with pytest.raises(native.GtaError):
        gta_amd.gta_attention(q, q, q, {"triv": 16}, {})
    with pytest.raises(NotImplementedError):
        gta_amd.Attention(32, 2, 16, attn_args={"method": {"name": "repast", "args": {}}})


def test_attention_kernel_is_spill_free():
    """The persistent attention kernel must compile without scratch for the shipped layouts (a spilled load sits behind
    a vmcnt(0) and serialises the prologue) and within 168 VGPRs at dh <= 64 (three workgroups per CU).
    tools/audit_spills.py compiles gta_fwd2.hip to assembly and counts."""
    import importlib.util
    import os
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_spills", os.path.join(root, "tools", "audit_spills.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    report, problems = mod.audit()
    assert report, "no gta_fwd2_kernel instantiation found"
    assert not problems, problems


def test_tuned_kernels_keep_their_register_budgets():
    """Pre-pass, backward kernels and the weight-gradient kernel's steady-state loop: no scratch where the measurements
    of DESIGN.md were taken without it (tools/audit_spills.py audit_others compiles the three files to assembly)."""
    import importlib.util
    import os
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_spills", os.path.join(root, "tools", "audit_spills.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    report, problems = mod.audit_others()
    assert len(report) >= 8 + 24 + 2, report
    assert not problems, problems


def test_attn64_loop_generator_checks_and_is_current():
    """gen_attn64.py emits the tile loop of gta_attn64_kernel and simulates it (typed dataflow of one wave over several tile
    counts, with forced rebase steps and a masked tail; LDS / DMA counters; the ISA's manual wait states).  The committed
    gta_attn64_loop.inc must be what the generator emits now."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gta_amd", "csrc", "gen_attn64.py")
    spec = importlib.util.spec_from_file_location("gen_attn64", path)
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    progs = {}
    for name, dh, kw in g.production_variants():
        gen, prog = g.build_program(dh, kw)               # (simulated inside)
        progs[name] = prog
        n_mfma = sum(1 for x in prog if x.kind == "mfma")
        assert n_mfma > (dh // 2) * 4
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "loop.inc")
        g.emit(progs, out)
        assert open(out).read() == open(os.path.join(root, "gta_amd", "csrc", "gta_attn64_loop.inc")).read(), \
            "gta_attn64_loop.inc is stale: python3 gta_amd/csrc/gen_attn64.py --out gta_amd/csrc/gta_attn64_loop.inc"
    # the simulation does catch what it is there for: a P.V MFMA moved in front of its fragment's wait, a dropped exp
    gen, prog = g.build_program(96, dict(g.production_variants()[0][2]))
    i = next(i for i, x in enumerate(prog) if x.kind == "wait" and x.sem[0] == "lgkm" and prog[i + 1].kind == "mfma")
    bad = prog[:i] + [prog[i + 1], prog[i]] + prog[i + 2:]
    with pytest.raises(g.CheckError):
        g.check_all(gen, bad)
    j = next(i for i, x in enumerate(prog) if x.sem and x.sem[0] == "exp" and any(l.label and l.label.startswith("L_step1") for l in prog[:i]))
    with pytest.raises(g.CheckError):
        g.check_all(gen, prog[:j] + prog[j + 1:])


def _load_csrc_module(name):
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gta_amd", "csrc")
    if d not in sys.path:
        sys.path.insert(0, d)          # (the generators import each other by plain module name)
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m, d


def test_item_stream_generator_simulates_and_is_current():
    """gen_item64.py emits the whole work item of gta_attn64_items_kernel (prologue, tile loop, epilogue, request for the next item) as
    one instruction stream.  Its check() EXECUTES the stream -- the tile loop replaced by a stub that verifies the prologue's results
    and leaves the loop's -- on the functional wave simulator of isa_model.py (64 lanes, LDS, global memory, in-order memory counters
    with poisoned load destinations, the ISA's manual wait states) for three launches (exactly representable inputs: results bit for
    bit; random inputs: within one bf16 step; a single-item launch), all four waves, and compares Q' fragments, |q'| bounds, stream
    pointers, O rows, LSE and the profile stamps with a numpy model of the C++ prologue / epilogue.  The assembler must accept the
    text, and the committed gta_attn64_items.inc must be what the generator emits now."""
    import os
    import tempfile
    g, d = _load_csrc_module("gen_item64")
    st = g.check()
    # 2 x 12 MFMAs in the prologue and in each of the two epilogue copies (the statement's last item has its own: no request behind it);
    # ~1 100 instructions per item besides the tile loop, ~200 of per-launch setup
    assert st["mfma"] == 72 and st["instructions"] < 1900, st
    full = g.ItemGen().program()
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        assert g.assemble_check(full)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "items.inc")
        g.emit(out, full, g.ItemGen().program(pad4=True), g.ItemGen(phases=True).program())
        assert open(out).read() == open(os.path.join(d, "gta_attn64_items.inc")).read(), \
            "gta_attn64_items.inc is stale: make -C gta_amd/csrc regen"


@pytest.mark.parametrize("fault", ["fragment_offset", "early_wait", "tile_order", "rotation_sign", "missing_mask", "lse_offset", "missing_pad",
                                   "store_base"])
def test_item_stream_simulation_catches_faults(fault):
    """the simulation objects to what it is there for: a wrong LDS offset, a mis-counted vmcnt at the item's start, swapped operand
    tiles, a rotation sign, a dropped exec mask, a wrong LSE address, a dropped wait-state pad, a wrong O base"""
    import os
    g, d = _load_csrc_module("gen_item64")
    src = open(os.path.join(d, "gen_item64.py")).read()
    old, new = {
        "fragment_offset": ("a.ds_read(128, QR[rb][ks], V_XFR, 32 * ks)", "a.ds_read(128, QR[rb][ks], V_XFR, 16 * ks)"),
        "early_wait": ("        a.waitcnt(vm=14)\n        self.phase(a, 0)", "        a.waitcnt(vm=20)\n        self.phase(a, 0)"),
        "tile_order": ("out.append((ACC[rb][d], AT[6 * lo + tl], QR[rb][tl], 0 if i < 3 else ACC[rb][d]))",
                       "out.append((ACC[rb][d], AT[6 * lo + (tl ^ 1)], QR[rb][tl], 0 if i < 3 else ACC[rb][d]))"),
        "rotation_sign": ("ops.append(lambda x0=x0, c=c, tt=tt: a.v_fma_f32(x0, x0, c, tt, neg=(False, False, True)))",
                          "ops.append(lambda x0=x0, c=c, tt=tt: a.v_fma_f32(x0, x0, c, tt))"),
        "missing_mask": ("                    if ks == 4:\n                        ops.append(lambda: self.hi(a))",
                         "                    if ks == 44:\n                        ops.append(lambda: self.hi(a))"),
        "lse_offset": ("a.global_store(1, V_LSE, [t[8 + rb]], cur(D_LSE), 128 * rb)", "a.global_store(1, V_LSE, [t[8 + rb]], cur(D_LSE), 64 * rb)"),
        "missing_pad": ("            if i == 2:\n                a.nop(7)", "            if i == 2:\n                a.nop(1)"),
        "store_base": ("S_B[2 * rb + i // 3]", "S_B[rb + i // 3]"),
    }[fault]
    assert old in src
    ns = {"__name__": "gen_item64_fault"}
    exec(compile(src.replace(old, new, 1), "gen_item64_fault.py", "exec"), ns)
    with pytest.raises(g.CheckError):
        ns["check"](waves=(1,))


def test_dkv64_stream_generator_simulates_and_is_current():
    """gen_bwd64.py emits the tile loop of gta_bwd_dkv64_kernel (dK/dV of the backward, 64 keys per wave) as one instruction stream.  Its
    check() EXECUTES the whole statement on the functional wave simulator of isa_model.py -- K'/V' fragment loads, the LDS-DMA ring (the
    harness verifies at every barrier that the wave's own pieces of the tile have landed, and stands in for the other three waves), every
    MFMA, transpose-read, exp2 and pack -- for 1, 2, 5 and 6 query tiles (the walk's entry, all four stage copies, every exit) and compares
    the 192 accumulator registers and the values handed to the epilogue with a numpy model; the ISA's manual wait states are checked on the
    executed order.  The assembler must accept the text, and the committed gta_bwd64_dkv.inc must be what the generator emits now."""
    import os
    import tempfile
    g, d = _load_csrc_module("gen_bwd64")
    st = g.check()
    assert st["mfma"] == 480 and st["instructions"] < 2700, st          # tile 0: 48; four stage copies of 96; the tail: 48
    prog = g.Gen().program()
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        assert g.assemble_check(prog)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dkv.inc")
        g.emit(out, prog)
        assert open(out).read() == open(os.path.join(d, "gta_bwd64_dkv.inc")).read(), "gta_bwd64_dkv.inc is stale: make -C gta_amd/csrc regen"


def test_dq64_stream_generator_simulates_and_is_current():
    """the same for the dQ walk of gta_bwd_dq64_kernel (gen_bwd64.py: GenDQ -- 64 query rows per wave, the Q'' / dO~ fragments stationary, the
    K' / V' tiles streamed): executed for 1, 2, 5 and 6 key tiles against a numpy model of dQ'^T, wait states on the executed order, the
    assembler, the committed gta_bwd64_dq.inc."""
    import os
    import tempfile
    g, d = _load_csrc_module("gen_bwd64")
    st = g.check_dq()
    assert st["mfma"] == 360 and st["instructions"] < 2100, st          # tile 0: 48; four stage copies of 72; the tail: 24
    prog = g.GenDQ().program()
    if os.path.exists("/opt/rocm/lib/llvm/bin/clang"):
        assert g.assemble_check(prog, g.Q_VOPS, g.Q_SOPS)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dq.inc")
        g.emit(out, prog, "GTA_BWD64_DQ", g.Q_VOPS, g.Q_SOPS, g.DQ_RESULTS)
        assert open(out).read() == open(os.path.join(d, "gta_bwd64_dq.inc")).read(), "gta_bwd64_dq.inc is stale: make -C gta_amd/csrc regen"


@pytest.mark.parametrize("fault", ["half_offset", "pair_operand", "seed", "early_wait"])
def test_dq64_stream_simulation_catches_faults(fault):
    """a key half's fragments read from the other half, a dQ MFMA fed with the pair's other transposed operand, the -D seed in place of
    -lse2, a DMA wait that lets a tile's pieces be late"""
    import os
    g, d = _load_csrc_module("gen_bwd64")
    src = open(os.path.join(d, "gen_bwd64.py")).read()
    old, new = {
        "half_offset": ("a.ds_read(128, slot[0:4], reg, base + hh * HALF + imm)", "a.ds_read(128, slot[0:4], reg, base + (1 - hh) * HALF + imm)"),
        "pair_operand": ("a.mfma(DQ[rb][d], slot[4 * i:4 * i + 4], se[\"e\"][rb][4 * t:4 * t + 4], DQ[rb][d])",
                         "a.mfma(DQ[rb][d], slot[4 * (1 - i):4 * (1 - i) + 4], se[\"e\"][rb][4 * t:4 * t + 4], DQ[rb][d])"),
        "seed": ("QINIT_L[rb] if ks == 0 else se[\"s\"][rb])", "QINIT_D[rb] if ks == 0 else se[\"s\"][rb])"),
        "early_wait": ("        a.waitcnt(vm=6)\n        a.barrier()", "        a.waitcnt(vm=8)\n        a.barrier()"),
    }[fault]
    assert old in src, fault
    ns = {"__name__": "gen_bwd64_fault"}
    exec(compile(src.replace(old, new, 1), "gen_bwd64_fault.py", "exec"), ns)
    with pytest.raises(g.CheckError):
        ns["check_dq"](cases=((5, 3),))


@pytest.mark.parametrize("fault", ["fragment_offset", "tr_offset", "early_wait", "operand_swap", "pack_order", "init_rows", "stage_reuse"])
def test_dkv64_stream_simulation_catches_faults(fault):
    """the simulation objects to what it is there for: a wrong fragment or transpose-read offset, a DMA wait that lets a tile's pieces be
    late, dK fed with P instead of dS, a bf16 pack that overwrites a value it still needs, statistics of the wrong rows, a request that
    overwrites the stage the walk still reads"""
    import os
    g, d = _load_csrc_module("gen_bwd64")
    src = open(os.path.join(d, "gen_bwd64.py")).read()
    old, new = {
        "fragment_offset": ("a.ds_read(128, slot[4:8], reg, base + IMG + qb * HALF + imm)", "a.ds_read(128, slot[4:8], reg, base + IMG + qb * HALF + imm + 16)"),
        "tr_offset": ("imm = base + qb * HALF + t * SL + (64 if d == 1 else 0)", "imm = base + qb * HALF + t * SL + (32 if d == 1 else 0)"),
        "early_wait": ("        a.waitcnt(vm=7)\n        a.barrier()", "        a.waitcnt(vm=9)\n        a.barrier()"),
        "operand_swap": ("a.mfma(DK[kb][d], slot[0:4], se[\"e\"][kb][4 * t:4 * t + 4], DK[kb][d])", "a.mfma(DK[kb][d], slot[0:4], se[\"s\"][kb][4 * t:4 * t + 4], DK[kb][d])"),
        "pack_order": ("ops.append(lambda a, i=i, s=s: a.v_cvt_pk_bf16_f32(s[i], s[2 * i], s[2 * i + 1]))", "ops.append(lambda a, i=i, s=s: a.v_cvt_pk_bf16_f32(s[7 - i], s[2 * i], s[2 * i + 1]))"),
        "init_rows": ("st * 512 + (32 * qb + 8 * g) * 4))", "st * 512 + (32 * qb + 8 * (g ^ 1)) * 4))"),
        "stage_reuse": ("dma = self.dma_ops((c + 2) % R)", "dma = self.dma_ops((c + 3) % R)"),
    }[fault]
    assert old in src, fault
    ns = {"__name__": "gen_bwd64_fault"}
    exec(compile(src.replace(old, new, 1), "gen_bwd64_fault.py", "exec"), ns)
    with pytest.raises(g.CheckError):
        ns["check"](cases=((5, 3),))


def test_dkv64_kernel_leaves_the_register_files_to_the_stream():
    """tools/audit_spills.py audit_dkv64: the generated statements of gta_bwd_dkv64_kernel / gta_bwd_dq64_kernel own the accumulator file and
    hand dK'^T / dV'^T (a[0:191]) / dQ'^T (a[0:95]) over as OUTPUTS of the statement; hipcc reads every result register once and does
    nothing else with the file; the statement's operands must have found room in v0..v23."""
    import importlib.util
    import os
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_spills", os.path.join(root, "tools", "audit_spills.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    report, problems = mod.audit_dkv64()
    # gta_bwd_dkv64_kernel, gta_bwd_dq64_kernel: one generated statement each; gta_bwd_dqkv64_kernel (both bodies in one launch): two
    assert sorted(r["loop_statements"] for r in report) == [1, 1, 2], report
    assert not problems, problems


def test_attn64_kernel_leaves_the_accumulator_file_to_the_loop_statement():
    """tools/audit_spills.py audit_attn64: gta_attn64_kernel's loop statement names what crosses its boundary in the accumulator file in its
    operand list (Q' fragments in, O out); one loop statement, 256 + 256 register split, (nearly) no scratch; the item stream's kernel keeps
    hipcc out of the accumulator file altogether."""
    import importlib.util
    import os
    import shutil
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("audit_spills", os.path.join(root, "tools", "audit_spills.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    report, problems = mod.audit_attn64()
    assert len(report) >= 5 and any(r["instance"] == "items" for r in report), report
    assert not problems, problems


def test_generated_streams_end_with_the_matrix_pipes_wait_states():
    """A stream whose accumulators are read by COMPILED code (outputs of the asm statement) must itself keep the XDL-write -> VALU-read distance
    of its last MFMAs: hipcc places no wait states behind inline asm.  r05 found the dQ walk ending on an MFMA (harmless while literal reads of
    a[0:15] .. a[80:95] followed in that order, wrong results in a[80:95] once hipcc picked the order).  18 states for a 16-pass MFMA."""
    import os
    import re
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gta_amd", "csrc")
    seen = 0
    for fname, macros in (("gta_bwd64_dkv.inc", ("GTA_BWD64_DKV",)), ("gta_bwd64_dq.inc", ("GTA_BWD64_DQ",)),
                          ("gta_attn64_loop.inc", ("GTA_ATTN64_LOOP_V0", "GTA_ATTN64_LOOP_V1", "GTA_ATTN64_LOOP64_V0", "GTA_ATTN64_LOOP64_V1"))):
        text = open(os.path.join(d, fname)).read()
        for mac in macros:
            i = text.index(f"#define {mac} \\\n")
            body = text[i:text.index('    ""\n', i)]
            ins = [m.group(1) for m in re.finditer(r'^\s+"([^"\\]+)\\n\\t"', body, re.M)]
            assert len(ins) > 500, (mac, len(ins))
            last = max(k for k, t in enumerate(ins) if t.startswith("v_mfma"))
            states = 0
            for t in ins[last + 1:]:
                m = re.match(r"s_nop (\d+)", t)
                states += int(m.group(1)) + 1 if m else (0 if t.endswith(":") else 1)
            assert states >= 18, (mac, states, ins[last:][:6])
            seen += 1
    assert seen == 6


def test_srt_wrapper_state_dict_is_reference_compatible():
    """gta_amd.srt.TransformingSRT takes the reference's cfg and loads the reference's own state dict
    (fixture srt_ms_tiny: parameters of the reference TransformingSRT) with strict=True."""
    import ast
    from tests import _golden as G
    from gta_amd import srt
    d, _ = G.load("srt_ms_tiny")
    cfg = ast.literal_eval(str(np.load(G.GOLDEN + "/srt_ms_tiny.npz")["meta"]))
    model = srt.TransformingSRT(cfg)
    sd = {k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")}
    assert list(sd) == [n for n, _ in model.named_parameters()]
    model.load_state_dict(sd, strict=True)
    with pytest.raises(NotImplementedError):
        bad = dict(cfg, encoder_kwargs=dict(cfg["encoder_kwargs"], emb="ray"))
        srt.TransformingSRT(bad)


def test_reference_checkpoint_layout_round_trip(tmp_path):
    """A file in the reference's checkpoint layout (checkpoint.py:21-35: one state dict per module + scalars),
    built from the reference's own SRT weights (fixture), loads into gta_amd.TransformingSRT strictly."""
    import ast
    from tests import _golden as G
    from gta_amd import srt, checkpoint
    d, _ = G.load("srt_ms_tiny")
    cfg = ast.literal_eval(str(np.load(G.GOLDEN + "/srt_ms_tiny.npz")["meta"]))
    sd = {k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")}
    ref_file = {"encoder": {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")},
                "decoder": {"module." + k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")},
                "epoch_it": 3, "it": 1234, "t": 56.0, "loss_val_best": 21.5, "run_id": "abc"}
    path = str(tmp_path / "model.pt")
    torch.save(ref_file, path)
    model = srt.TransformingSRT(cfg)
    rest = checkpoint.load_checkpoint(path, encoder=model.encoder, decoder=model.decoder)
    assert rest == {"epoch_it": 3, "it": 1234, "t": 56.0, "loss_val_best": 21.5, "run_id": "abc"}
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), sd[n]), n
    out = str(tmp_path / "again.pt")
    checkpoint.save_checkpoint(out, {"it": 1}, encoder=model.encoder, decoder=model.decoder)
    again = torch.load(out, weights_only=False)
    assert sorted(again) == ["decoder", "encoder", "it"] and list(again["encoder"]) == list(ref_file["encoder"])


def test_checkpoint_written_by_the_reference_loads_strictly():
    """``tests/golden/ckpt_ref_ms.pt`` was written by the reference's own ``Checkpoint.save`` (checkpoint.py:21-35) with encoder,
    decoder and optimizer registered and train.py:301-305's scalars (oracle/make_golden.py ``checkpoint_case``); ``ckpt_ref_ms_ddp.pt`` carries the
    ``module.`` prefix of wrapped modules.  Both load into gta_amd.TransformingSRT with ``strict=True`` under ``weights_only=True``, the optimizer
    entry restores an AdamW over the same parameters, the scalars come back as ``Checkpoint.load`` returns them, and the oracle model under the
    loaded weights reproduces the reference's rendering of the fixture batch (CPU leg; the HIP leg is tests/test_gpu_modules.py)."""
    import ast
    from tests import _golden as G
    from gta_amd import srt, checkpoint
    from oracle import gta_oracle as O
    io = np.load(G.GOLDEN + "/ckpt_ref_ms_io.npz")
    cfg = ast.literal_eval(str(io["meta"]))
    scalars = ast.literal_eval(str(io["scalars"]))
    model = srt.TransformingSRT(cfg)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    rest = checkpoint.load_checkpoint(G.GOLDEN + "/ckpt_ref_ms.pt", encoder=model.encoder, decoder=model.decoder, optimizer=opt)
    assert rest == scalars
    st = opt.state_dict()["state"]
    assert len(st) == len(list(model.parameters())) and all(float(v["step"]) == 1.0 for v in st.values())
    raw = torch.load(G.GOLDEN + "/ckpt_ref_ms.pt", weights_only=True)
    for part in ("encoder", "decoder"):
        for n, v in getattr(model, part).state_dict().items():
            assert torch.equal(v, raw[part][n]), (part, n)
    ddp = srt.TransformingSRT(cfg)
    rest = checkpoint.load_checkpoint(G.GOLDEN + "/ckpt_ref_ms_ddp.pt", encoder=ddp.encoder, decoder=ddp.decoder)
    assert rest == scalars
    for (n, a), (_, b) in zip(model.state_dict().items(), ddp.state_dict().items()):
        assert torch.equal(a, b), n
    with pytest.raises(KeyError):
        checkpoint.load_checkpoint(G.GOLDEN + "/ckpt_ref_ms_ddp.pt", optimizer=opt)
    om = O.OracleSRT(cfg).double().eval()
    om.load_state_dict({k: v.double() for k, v in model.state_dict().items()}, strict=True)
    t = lambda n: torch.from_numpy(io[n]).double()
    ex = {k[len("extras."):]: torch.from_numpy(io[k]).double() for k in io.files if k.startswith("extras.")}
    with torch.no_grad():
        pred = om(t("images"), t("cam_in"), t("rays_in"), t("cam_t"), t("rays_t"), ex)
    assert (pred - t("pred")).abs().max() < 1e-9


def test_j_convention_check(tmp_path):
    """a J_dense.pt is compared with the build's J before a released so3 checkpoint is served.  The real
    blob is not in the image, so the check itself is tested: identical J -> trivial signs; a sign-conjugated J (another
    real-SH sign convention) -> the conjugating S is found and the oracle's Wigner-D under that J is S D S; anything
    else raises.  The build's constants equal the oracle's (which make_golden.py uses as the stand-in blob)."""
    from gta_amd import checkpoint
    from oracle import gta_oracle as O
    for l, mine in ((1, checkpoint.J1), (2, checkpoint.J2)):
        assert torch.allclose(torch.tensor(mine, dtype=torch.float64), O.J_MATRICES[l], atol=1e-15)
    blob = [O.J_MATRICES[l].clone() for l in range(3)]
    path = str(tmp_path / "J_dense.pt")
    torch.save(blob, path)
    S = checkpoint.check_j_convention(path)
    assert all((S[l] == 1).all() for l in (1, 2))
    s1 = torch.tensor([-1.0, 1.0, -1.0], dtype=torch.float64)
    s2 = torch.tensor([-1.0, 1.0, 1.0, 1.0, -1.0], dtype=torch.float64)
    flipped = [blob[0], s1[:, None] * blob[1] * s1[None], s2[:, None] * blob[2] * s2[None]]
    S = checkpoint.check_j_convention(flipped)
    assert torch.equal(S[1], s1) and torch.equal(S[2], s2)
    # what that means for the reps: D under the flipped J is S D S
    g = torch.Generator().manual_seed(0)
    R = O.random_extrinsics(1, 4, g, torch.float64)[0, :, :3, :3]
    D = O.wigner_d_euler(2, R)
    saved = {l: O.J_MATRICES[l] for l in (1, 2)}
    try:
        O.J_MATRICES[1], O.J_MATRICES[2] = flipped[1], flipped[2]
        Df = O.wigner_d_euler(2, R)
    finally:
        O.J_MATRICES[1], O.J_MATRICES[2] = saved[1], saved[2]
    assert (Df[1] - s1[:, None] * D[1] * s1[None]).abs().max() < 1e-12
    assert (Df[2] - s2[:, None] * D[2] * s2[None]).abs().max() < 1e-12
    with pytest.raises(ValueError):
        checkpoint.check_j_convention([blob[0], torch.eye(3, dtype=torch.float64), blob[2]])
    with pytest.raises(ValueError):
        checkpoint.check_j_convention([blob[0], blob[1], blob[2].roll(1, 0)])


def test_attention_kernel_selection_and_backward_workspace_without_gpu():
    """Host logic of the C ABI that needs no device: which attention kernel a descriptor gets (gta_debug_attention_kernel: the 64-rows-per-wave
    kernel for dh = 96 in the MSN layout and dh = 64 bf16 in the CLEVR-TR / pure-so2 layouts when the key side is whole ring turns of 64-key
    tiles and there are more than 128 query rows), and the size of the backward's workspace."""
    MS, CL, DIT = {"se3": 48, "so3": 24, "so2": 24}, {"se3": 32, "so2": 32}, {"so2": 64}

    def kern(dh, f, L, Tq, Tk, dtype, N=1, flags=0):
        q = torch.empty(2, 4, Tq, dh, dtype=dtype)
        k = torch.empty(2, 4, Tk, dh, dtype=dtype)
        d = native.make_desc(q, k, k, q, f, L, N, N, dh ** -0.5, native.FLAG_V_TRANSFORM | flags)
        return native.attention_kernel(d)[0::2]
    bf, f32 = torch.bfloat16, torch.float32
    assert kern(96, MS, 2, 1280, 1280, bf, 5) == ("gta_attn64_items_kernel", 256)        # whole one-view items, bf16: the generated item stream
    assert kern(96, MS, 2, 1280, 1280, bf, 5, native.FLAG_ITEM_CXX) == ("gta_attn64_kernel", 256)
    assert kern(96, MS, 2, 1536, 1536, bf, 12) == ("gta_attn64_kernel", 256)             # 128-token views: items span views
    assert kern(96, MS, 2, 1280, 1280, f32, 5) == ("gta_attn64_kernel", 256)
    assert kern(96, MS, 2, 1280, 1280, bf, 5, native.FLAG_ROWS32) == ("gta_fwd2_kernel", 128)
    assert kern(96, MS, 2, 1280, 640, bf, 5) == ("gta_fwd2_kernel", 128)            # 10 key tiles: not whole ring turns
    assert kern(96, MS, 2, 125, 1280, bf, 5) == ("gta_fwd2_kernel", 128)            # no more than 128 query rows: the 32-row kernel
    assert kern(64, CL, 0, 512, 512, bf, 2) == ("gta_attn64_kernel", 256)
    assert kern(64, CL, 0, 512, 512, f32, 2) == ("gta_fwd2_kernel", 128)            # the dh = 64 instances are bf16
    assert kern(64, CL, 0, 600, 600, bf, 2) == ("gta_fwdc_kernel", 128)             # CLEVR-TR's own Tk = 600: 9.4 key tiles -> the dh = 64 bf16 instance (r06)
    assert kern(64, CL, 0, 600, 600, bf, 2, native.FLAG_FWD2_GENERIC) == ("gta_fwd2_kernel", 128)
    assert kern(64, CL, 0, 600, 64, bf, 2) == ("gta_fwd2_kernel", 128)              # one key tile: the generic kernel (bit-for-bit with the single-kernel plan)
    assert kern(64, DIT, 0, 1024, 1024, bf) == ("gta_attn64_kernel", 256)
    q = torch.empty(2, 8, 1280, 96, dtype=bf)
    d0 = native.make_desc(q, q, q, q, MS, 2, 5, 5, 96 ** -0.5, native.FLAG_V_TRANSFORM)
    w0 = native.lib().gta_attn_bwd_workspace_bytes(ctypes.byref(d0))
    img = 2 * 8 * 20 * 2 * 64 * 96 * 2                                                  # Q''/dO~ (and recomputed K'/V') tile images
    assert 2 * img < w0 < 2 * img + 2 * 8 * 20 * 128 * 4 + 64 * 1024                     # + per-row statistics and the partial sums


def test_bench_host_helpers():
    """bench.py's host-side helpers (r05): the container's CPU quota is read from the cgroup and bounds the thread counts the timed legs use;
    `precondition` runs its callable for the asked time in whole chunks; the workload table covers the five BASELINE workloads."""
    import importlib.util
    import os
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    q = bench.cpu_quota()
    assert 1 <= q <= (os.cpu_count() or 1)
    assert 1 <= bench.HOST_THREADS <= min(8, q)
    calls = []
    t0 = time.perf_counter()
    n = bench.precondition(lambda: calls.append(1), 0.05, chunk=7)
    assert n == len(calls) and n % 7 == 0 and n >= 7 and time.perf_counter() - t0 >= 0.05
    assert bench.precondition(lambda: calls.append(1), 0.0) == 0
    # the five BASELINE workloads (the default line times exactly these) + the `--workload`-only MSN layouts without the so3 slab (r06)
    assert set(bench.WORKLOADS) == {"ms-enc", "ms-dec", "cl-enc", "cl-dec", "dit", "ms-gta-enc", "ms-gta-dec", "ms-se3-enc"}
    assert os.environ.get("OMP_WAIT_POLICY") == "PASSIVE"
