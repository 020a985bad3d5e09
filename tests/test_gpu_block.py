"""GPU: the fused Transformer block (include/gta_block.h, gta_amd.fused).

Row kernels and GEMM epilogues against plain PyTorch fp32 references of the same operations (nn.LayerNorm,
nn.GELU, nn.Linear: source/layers.py:146-169,388-395,429-430), then the fused ``Transformer`` against the
module-by-module path under identical weights, forward and every gradient.  The reference-generated module
fixtures (tests/golden/mod_*.npz, srt_ms_tiny.npz) run through the fused path in test_gpu_modules.py.
"""
import pytest
import torch
import torch.nn.functional as F

import gta_amd
from gta_amd import fused, layers, synth
from gta_amd.native import GtaError
from gta_amd import native_block as nb
from tests import _hip_cases as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _err(a, b):
    return (a.float() - b.float()).abs().max().item()


@pytest.mark.parametrize("rows,d", [(37, 768), (4, 8), (129, 512), (50, 1024), (33, 1536), (9, 2048), (5, 4096), (1, 264),
                                    (41, 180), (6, 4), (17, 1020), (3, 2044)])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, BF), (torch.float32, torch.float32), (BF, BF)])
def test_layernorm_forward_backward(rows, d, xdt, ydt):
    g = torch.Generator(device=DEV).manual_seed(rows * 7 + d)
    x = (torch.randn(rows, d, device=DEV, generator=g) * 1.7 + 0.4).to(xdt)
    gam = torch.randn(d, device=DEV, generator=g) * 0.3 + 1.0
    bet = torch.randn(d, device=DEV, generator=g) * 0.2
    dy = torch.randn(rows, d, device=DEV, generator=g).to(ydt)
    dres = torch.randn(rows, d, device=DEV, generator=g).to(xdt)
    y, mean, rstd = nb.ln_fwd(x, gam, bet, 1e-5, ydt)
    xr = x.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = F.layer_norm(xr, (d,), gr, br, 1e-5)
    ref.backward(dy.double())
    tol_y = 2.0 ** -8 * ref.abs().max().item() if ydt == BF else 2e-6 * max(1.0, ref.abs().max().item())
    assert _err(y, ref) <= tol_y
    assert _err(mean, xr.detach().mean(1)) < 1e-5 and _err(rstd * xr.detach().var(1, unbiased=False).add(1e-5).sqrt(), torch.ones(rows, device=DEV)) < 1e-5
    dx, dgam, dbet = nb.ln_bwd(dy, x, gam, mean, rstd, dres)
    want = xr.grad + dres.double()
    tol_dx = 2.0 ** -8 * want.abs().max().item() if xdt == BF else 1e-5 * max(1.0, want.abs().max().item())
    assert _err(dx, want) <= tol_dx
    assert _err(dgam, gr.grad) <= 1e-5 * max(1.0, gr.grad.abs().max().item()) * rows ** 0.5
    assert _err(dbet, br.grad) <= 1e-5 * max(1.0, br.grad.abs().max().item()) * rows ** 0.5
    dx0, _, _ = nb.ln_bwd(dy, x, gam, mean, rstd, None)                       # no skip gradient
    assert _err(dx0, xr.grad) <= tol_dx
    a = nb.ln_bwd(dy, x, gam, mean, rstd, dres)                                # deterministic
    assert torch.equal(a[1], dgam) and torch.equal(a[2], dbet) and torch.equal(a[0], dx)


def test_layernorm_refuses_bad_shapes():
    x = torch.randn(4, 10, device=DEV)
    with pytest.raises(GtaError):
        nb.ln_fwd(x, torch.ones(10, device=DEV), torch.zeros(10, device=DEV), 1e-5, BF)
    with pytest.raises(GtaError):
        nb.ln_fwd(torch.randn(2, 2052, device=DEV), torch.ones(2052, device=DEV), torch.zeros(2052, device=DEV), 1e-5, BF)
    with pytest.raises(GtaError):
        nb.ln_fwd(torch.randn(4, 16), torch.ones(16), torch.zeros(16), 1e-5, BF)      # CPU tensors: no fallback


def test_bindings_refuse_malformed_operands():
    """The kernels index by the sizes they are told: the bindings check what the pointers stand for."""
    x = torch.randn(8, 64, device=DEV)
    g, b = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    y, mean, rstd = nb.ln_fwd(x, g, b, 1e-5, BF)
    with pytest.raises(GtaError):
        nb.ln_fwd(x, g.to(BF), b, 1e-5, BF)                       # gamma is read as fp32
    with pytest.raises(GtaError):
        nb.ln_fwd(x, g[:32], b, 1e-5, BF)                          # too short
    with pytest.raises(GtaError):
        nb.ln_fwd(x.t().contiguous().t(), g, b, 1e-5, BF)          # rows not contiguous
    with pytest.raises(GtaError):
        nb.ln_bwd(y[:4], x, g, mean, rstd, None)                   # gradient of another shape
    with pytest.raises(GtaError):
        nb.ln_bwd(y, x, g, mean[:4], rstd, None)                   # statistics of another row count
    with pytest.raises(GtaError):
        nb.ln_bwd(y, x, g, mean, rstd, x.to(BF))                   # skip gradient must have x's dtype
    with pytest.raises(GtaError):
        nb.gelu_bwd(y.float(), y)                                  # dtypes differ
    a, w = torch.randn(16, 64, device=DEV, dtype=BF), torch.randn(32, 64, device=DEV, dtype=BF)
    with pytest.raises(GtaError):
        nb.gemm(a, w, trans_b=True, epilogue=nb.EPI_BIAS, bias=torch.zeros(16, device=DEV, dtype=BF))   # bias of the wrong length
    with pytest.raises(GtaError):
        nb.gemm(a, w, trans_b=True, c=torch.zeros(16, 16, device=DEV, dtype=BF), beta=1.0)              # C of the wrong shape
    with pytest.raises(GtaError):
        nb.wgrad(torch.randn(64, 100, device=DEV, dtype=BF), torch.randn(64, 256, device=DEV, dtype=BF), False)   # n % 256 != 0


@pytest.mark.parametrize("dt", [torch.float32, BF])
def test_gelu_and_colsum(dt):
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(301, 1032, device=DEV, generator=g) * 2).to(dt)
    dy = torch.randn(301, 1032, device=DEV, generator=g).to(dt)
    xr = x.double().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(dy.double())
    tol = 2.0 ** -8 if dt == BF else 2e-6
    assert _err(nb.gelu_fwd(x), yr) <= tol * yr.abs().max().item()
    assert _err(nb.gelu_bwd(dy, x), xr.grad) <= tol * xr.grad.abs().max().item()
    want = x.double().sum(0)
    assert _err(nb.colsum(x), want) <= 1e-5 * x.double().abs().sum(0).max().item()
    assert _err(nb.colsum(x[:, 8:520]), want[8:520]) <= 1e-5 * x.double().abs().sum(0).max().item()   # strided rows
    odd = torch.randn(333, 180, device=DEV, generator=g).to(dt)                  # rows of 4k elements: 8-byte accesses
    assert _err(nb.colsum(odd), odd.double().sum(0)) <= 1e-5 * odd.double().abs().sum(0).max().item()
    big = torch.randn(5000, 3072, device=DEV, generator=g).to(dt)
    assert _err(nb.colsum(big), big.double().sum(0)) <= 1e-5 * big.double().abs().sum(0).max().item()
    assert torch.equal(nb.colsum(big), nb.colsum(big))


@pytest.mark.parametrize("cdt", [BF, torch.float32])
def test_gemm_epilogues(cdt):
    """Every form the block uses, row-major, against fp64 matmul."""
    g = torch.Generator(device=DEV).manual_seed(11)
    M, K, N = 200, 72, 136
    a = torch.randn(M, K, device=DEV, generator=g).to(cdt)
    W = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).to(cdt)
    b = torch.randn(N, device=DEV, generator=g) * 0.3
    skip = torch.randn(M, N, device=DEV, generator=g)
    ref = a.double() @ W.double().t()
    tol = (2.0 ** -8 if cdt == BF else 1e-5) * ref.abs().max().item()
    assert _err(nb.gemm(a, W, trans_b=True), ref) <= tol                                             # x W^T
    assert _err(nb.gemm(a, W, trans_b=True, epilogue=nb.EPI_BIAS, bias=b), ref + b) <= 1.5 * tol      # + bias
    y = nb.gemm(a, W, trans_b=True, epilogue=nb.EPI_BIAS, bias=b, c=skip, beta=1.0, out_dtype=torch.float32)
    assert y.dtype == torch.float32 and _err(y, ref + b + skip) <= 1e-5 * ref.abs().max().item() + 1e-5   # + skip, fp32 out
    h = nb.gemm(a, W, trans_b=True, epilogue=nb.EPI_BIAS_GELU, bias=b)                                # tanh-form GELU
    assert _err(h, F.gelu(ref + b, approximate="tanh")) <= 1.5 * tol
    assert _err(h, F.gelu(ref + b)) <= 1.5 * tol + 5e-4
    dout = torch.randn(M, N, device=DEV, generator=g).to(cdt)
    assert _err(nb.gemm(dout, W), dout.double() @ W.double()) <= tol * 4                              # dgrad
    dW = nb.gemm(dout, a, trans_a=True, out_dtype=torch.float32)                                      # wgrad, fp32 out
    want = dout.double().t() @ a.double()
    assert dW.dtype == torch.float32 and _err(dW, want) <= 1e-5 * want.abs().max().item() * (30 if cdt == BF else 1)
    with pytest.raises(GtaError):
        nb.gemm(a, W)                                                                                 # inner dimensions differ
    if cdt == BF:
        with pytest.raises(GtaError):       # hipBLASLt reads a bf16 bias as garbage when D is fp32: the ABI refuses the pair
            nb.gemm(a, W, trans_b=True, epilogue=nb.EPI_BIAS, bias=b.to(BF), out_dtype=torch.float32)


def _transformer(cross, seed=0, dim=128):
    torch.manual_seed(seed)
    f_dims = {"triv": 0, "se3": 32, "so3": 0, "so2": 32}
    ak = {"f_dims": f_dims, "so2": 8, "so3": 0, "max_freq_h": 1, "max_freq_w": 1}
    tr = gta_amd.Transformer(dim, 2, 2, 64, 2 * dim, 0.0, not cross, 96 if cross else None, False,
                             {"method": {"name": "gta", "args": ak}}).to(DEV)
    for n, p in tr.named_parameters():                      # biases and norms away from their trivial initial values
        if n.endswith("bias") or "norm" in n:
            with torch.no_grad():
                p.add_(torch.randn_like(p) * 0.1)
    B, V, hw = 2, 3, 6
    gen = torch.Generator().manual_seed(3)
    ex = {"input_transforms": synth.random_extrinsics(B, V, gen).to(DEV),
          "input_coord": torch.rand(B, V, hw, hw, 2, generator=gen).to(DEV)}
    gta_amd.pre_compute_reps_encoder(ak, ex)
    if cross:
        ex["target_transforms"] = synth.random_extrinsics(B, 1, gen).to(DEV)
        ex["target_coord"] = torch.rand(B, 40, 2, generator=gen).to(DEV)
        gta_amd.pre_compute_reps_decoder(ak, ex)
    x = torch.randn(B, 40 if cross else V * hw * hw, dim, device=DEV)
    z = torch.randn(B, V * hw * hw, 96, device=DEV) if cross else None
    return tr, ex, x, z


def _run(tr, ex, x, z, fused_on, autocast):
    tr.fused_blocks = fused_on
    try:
        for p in tr.parameters():
            p.grad = None
        x = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=BF, enabled=autocast):
            y = tr(x, z, ex)
        w = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        (y.float() * w).sum().backward()
        torch.cuda.synchronize()
        return y.detach().float(), x.grad.clone(), {n: p.grad.clone() for n, p in tr.named_parameters()}
    finally:
        tr.fused_blocks = True


@pytest.mark.parametrize("cross,dim", [(False, 128), (True, 128), (True, 180), (False, 180)])
@pytest.mark.parametrize("autocast", [False, True])
def test_fused_transformer_matches_modulewise(cross, dim, autocast):
    """Same weights, same inputs: fused blocks vs LayerNorm / Linear / GELU modules + autograd, forward and all gradients.
    dim = 180: the MSN decoder's width (rows of 4k elements: the 8-byte-access forms of the row kernels, hipBLASLt on
    leading dimensions that are not 16-byte multiples, transposed-GEMM weight gradients)."""
    tr, ex, x, z = _transformer(cross, dim=dim)
    spy = []
    orig_ln = nb.ln_fwd
    nb.ln_fwd = lambda *a, **k: (spy.append(1), orig_ln(*a, **k))[1]
    y0, dx0, g0 = _run(tr, ex, x, z, False, False)                 # fp32 module-by-module: the yardstick
    try:
        y1, dx1, g1 = _run(tr, ex, x, z, True, autocast)
    finally:
        nb.ln_fwd = orig_ln
    assert len(spy) == 4                                           # the fused path ran (two LayerNorm kernels per layer)
    rel = 3e-2 if autocast else 3e-3       # fp32: same arithmetic; only summation orders and the bf16 roundings they flip in the attention kernel differ
    st = C.err_stats(y1.cpu(), y0.cpu())
    assert st["finite"] and st["rel_rms"] < rel, st
    st = C.err_stats(dx1.cpu(), dx0.cpu())
    assert st["finite"] and st["rel_rms"] < 2 * rel, st
    for n in g0:
        st = C.err_stats(g1[n].float().cpu(), g0[n].cpu())
        assert st["finite"], (n, st)
        if n.endswith("trans_coeff"):
            continue                                               # cancellation-dominated scalar (operator tests)
        assert st["max_abs"] <= (8e-2 if autocast else 4e-2) * max(st["ref_max"], 1e-3) + 1e-5, (n, st)


def test_fused_path_is_taken_and_launch_count():
    """The fused layer issues 8 kernels beside the attention operator's; the module path under autocast many more."""
    tr, ex, x, z = _transformer(False)
    calls = []
    orig = {n: getattr(nb, n) for n in ("ln_fwd", "gemm", "gelu_fwd")}
    for n, f in orig.items():
        setattr(nb, n, (lambda n_, f_: lambda *a, **k: (calls.append(n_), f_(*a, **k))[1])(n, f))
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=BF):
            tr(x, z, ex)
    finally:
        for n, f in orig.items():
            setattr(nb, n, f)
    per_layer = ["ln_fwd", "gemm", "gemm", "ln_fwd", "gemm", "gemm"]     # bf16 inference: GELU in the epilogue
    assert calls == per_layer * 2, calls


def test_inference_gelu_epilogue_close_to_exact():
    tr, ex, x, z = _transformer(False)
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        y_epi = tr(x, z, ex).float()
        fused.GELU_EPILOGUE = False
        try:
            y_exact = tr(x, z, ex).float()
        finally:
            fused.GELU_EPILOGUE = True
    assert C.err_stats(y_epi.cpu(), y_exact.cpu())["rel_rms"] < 3e-3


def test_weight_cast_cache_follows_updates():
    """The bf16 copies of fp32 weights follow every kind of update -- also one made through ``.data`` (EMA, clamp, copy_), which
    changes neither the parameter's version counter nor its storage address (the default: a fresh cast per call)."""
    from gta_amd import fused
    tr, ex, x, z = _transformer(False)
    assert fused.CACHE_WEIGHT_CASTS is False
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        y0 = tr(x, z, ex).float()
        w = tr.layers[0][1].fn.net[0].weight
        w.mul_(1.5)                                   # in-place update of the parameter
        y1 = tr(x, z, ex).float()
        w.data.mul_(1.0 / 1.5)                        # update through .data: no version bump, same storage
        y2 = tr(x, z, ex).float()
    assert (y1 - y0).abs().max() > 1e-3
    assert (y2 - y0).abs().max() < 1e-6 * (1 + y0.abs().max())
    # the opt-in cache keeps its documented contract (version counter / storage address) and can be dropped by hand
    fused.CACHE_WEIGHT_CASTS = True
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=BF):
            tr(x, z, ex)
            w.mul_(1.5)
            y3 = tr(x, z, ex).float()
            assert (y3 - y1).abs().max() < 1e-6 * (1 + y1.abs().max())
            w.data.mul_(1.0 / 1.5)
            fused.clear_weight_casts(tr)
            y4 = tr(x, z, ex).float()
            assert (y4 - y0).abs().max() < 1e-6 * (1 + y0.abs().max())
    finally:
        fused.CACHE_WEIGHT_CASTS = False
        fused.clear_weight_casts(tr)


def test_packed_projection_gradient_matches_separate_tensors():
    """dq, dk, dv written by the attention backward as slices of one packed buffer (the gradient of ``to_qkv``'s output)
    equal the gradients of separately stored q, k, v; same for the packed K/V projection of cross-attention."""
    from gta_amd import gta as G
    f_dims = {"triv": 0, "se3": 32, "so3": 0, "so2": 32}
    q, k, v, ex, ak, cross = C.synth_inputs(2, 2, 3, 40, 3, 40, f_dims, 8, 0, torch.float32, seed=2)
    ex = {n: t.to(DEV) for n, t in ex.items()}
    gta_amd.pre_compute_reps_encoder(ak, ex)
    packed_reps = G.pack_reps(ex, f_dims)
    tc = torch.tensor([0.01], device=DEV)
    B, H, T, dh = q.shape
    w = torch.randn(B, H, T, dh, device=DEV)

    def run(qq, kk, vv):
        out = G.gta_attention(qq, kk, vv, f_dims, packed_reps, so3_degree=0, trans_coeff=tc, scale=dh ** -0.5)
        (out.float() * w).sum().backward()

    for dt in (torch.float32, BF):
        leaves = [t.to(DEV).to(dt).clone().requires_grad_(True) for t in (q, k, v)]
        run(*leaves)
        qkv = torch.stack([t.detach().permute(0, 2, 1, 3) for t in leaves], dim=2).contiguous().requires_grad_(True)   # [B,T,3,H,dh]
        run(*G.split_packed(qkv))
        same = lambda a, b: C.err_stats(a.float(), b.float())["rel_rms"] < (2e-3 if dt == BF else 2e-5)   # the kernels pick
        # their plan from the strides, so the two layouts may differ in summation order, not in values
        for i, t in enumerate(leaves):
            assert same(qkv.grad[:, :, i].permute(0, 2, 1, 3), t.grad), (dt, i)
        # packed K/V only (cross-attention: layers.py:394-395)
        ql = leaves[0].detach().clone().requires_grad_(True)
        kv = torch.stack([t.detach().permute(0, 2, 1, 3) for t in leaves[1:]], dim=2).contiguous().requires_grad_(True)
        run(ql, *G.split_packed(kv))
        assert same(ql.grad, leaves[0].grad)
        for i, t in enumerate(leaves[1:]):
            assert same(kv.grad[:, :, i].permute(0, 2, 1, 3), t.grad), (dt, "kv", i)


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (1280, 512, 256), (4160, 768, 256), (64, 256, 512), (40960, 768, 768)])
def test_wgrad_kernel(m, n, k):
    """The hand-written TN weight-gradient kernel (gta_wgrad) against fp64, with and without the fused bias gradient,
    on strided operands (slices of wider matrices, as the packed projections are)."""
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    G = torch.randn(m, n + 64, device=DEV, generator=g).to(BF)[:, 32:32 + n]
    X = torch.randn(m, k + 8, device=DEV, generator=g).to(BF)[:, :k]
    assert nb.wgrad_supported(G, X)
    want = G.double().t() @ X.double()
    wb = G.double().sum(0)
    scale = (m ** 0.5)
    dw, db = nb.wgrad(G, X, True)
    assert dw.dtype == torch.float32 and _err(dw, want) <= 2e-5 * scale * 6
    assert _err(db, wb) <= 2e-5 * scale * 6
    dw2, none = nb.wgrad(G, X, False)
    assert none is None and torch.equal(dw2, dw)                      # deterministic, bias leg changes nothing
    assert not nb.wgrad_supported(G[:-1], X[:-1]) and not nb.wgrad_supported(G[:, :128], X)


def test_bf16_side_copy_of_the_stream_gradient():
    """gta_ln_bwd's second output: dx in bf16 beside the fp32 result, consumed by the block upstream instead of a cast;
    a copy whose tensor was modified in place afterwards (autograd accumulating into it) must not be used."""
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(64, 256, device=DEV, generator=g)
    gam, bet = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    y, mean, rstd = nb.ln_fwd(x, gam, bet, 1e-5, BF)
    dy = torch.randn(64, 256, device=DEV, generator=g).to(BF)
    dx, _, _ = nb.ln_bwd(dy, x, gam, mean, rstd, None, bf16_copy=True)
    side, ver = dx._gta_bf16
    assert side.dtype == BF and torch.equal(side, dx.to(BF)) and ver == dx._version
    shaped = fused._shaped(dx, (4, 16, 256))
    assert fused._in_compute_dtype(shaped.reshape(-1, 256), shaped, BF).data_ptr() == side.data_ptr()
    shaped.add_(1.0)                                              # what an in-place gradient accumulation does
    fresh = fused._in_compute_dtype(shaped.reshape(-1, 256), shaped, BF)
    assert fresh.data_ptr() != side.data_ptr() and torch.equal(fresh, shaped.reshape(-1, 256).to(BF))


def _mask(shape, p, seed, dt=torch.float32):
    """The keep factors (0 or 1/(1-p)) of a dropout call, recovered through the kernel itself."""
    return nb.dropout_add(torch.ones(shape, device=DEV, dtype=dt), torch.zeros(shape, device=DEV, dtype=dt), p, seed)


def test_dropout_kernels():
    n = (1 << 20)
    for p in (0.01, 0.3):
        m = _mask((n,), p, 1234)
        kept = (m > 0).float().mean().item()
        assert abs(kept - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, (p, kept)
        assert torch.all((m == 0) | ((m - 1 / (1 - p)).abs() < 1e-6))
        assert torch.equal(m, _mask((n,), p, 1234)) and not torch.equal(m, _mask((n,), p, 1235))
        x = torch.randn(n, device=DEV)
        assert torch.allclose(nb.dropout_bwd(x, torch.float32, p, 1234), x * m)
        assert torch.equal(nb.dropout_bwd(x, BF, p, 1234), (x * m).to(BF))
        assert torch.allclose(nb.gelu_fwd(x, p, 1234), F.gelu(x) * m, atol=1e-6)
        xr = x.clone().requires_grad_(True)
        (F.gelu(xr) * m).backward(torch.ones_like(x) * 0.5)
        assert torch.allclose(nb.gelu_bwd(torch.full_like(x, 0.5), x, p, 1234), xr.grad, atol=1e-6)
        skip = torch.randn(n, device=DEV)
        assert torch.allclose(nb.dropout_add(x.to(BF), skip, p, 1234), skip + x.to(BF).float() * m, atol=1e-6)
    assert torch.equal(nb.gelu_fwd(x), nb.gelu_fwd(x, 0.0, 99))                 # p = 0: no mask, the seed is irrelevant
    # adjacent masks are unrelated (counter-based generator: no visible structure at stride 8)
    m = _mask((n,), 0.5, 7).view(-1, 8)
    assert abs(((m[:-1] > 0) == (m[1:] > 0)).float().mean().item() - 0.5) < 0.01


@pytest.mark.parametrize("autocast", [False, True])
def test_fused_functions_with_dropout_match_masked_reference(autocast):
    """to_out + Dropout + skip and PreNorm(FeedForward with its two Dropouts) + skip with fixed seeds against plain PyTorch
    using the same masks (recovered from the kernels), forward and all gradients."""
    torch.manual_seed(0)
    cdt = BF if autocast else torch.float32
    M, D, Fh, p, seed = 192, 128, 256, 0.25, 4242
    x = torch.randn(2, M // 2, D, device=DEV)
    norm = torch.nn.LayerNorm(D).to(DEV)
    l1, l2, lo = layers.ViTLinear(D, Fh).to(DEV), layers.ViTLinear(Fh, D).to(DEV), layers.JaxLinear(D, D).to(DEV)
    with torch.no_grad():
        for m in (norm, l1, l2, lo):
            for q in m.parameters():
                q.add_(torch.randn_like(q) * 0.05)
    a = torch.randn(2, M // 2, D, device=DEV).to(cdt)
    w = torch.randn(2, M // 2, D, device=DEV)
    params = [q for m in (norm, l1, l2, lo) for q in m.parameters()]

    def grads(out, xin):
        for q in params:
            q.grad = None
        (out.float() * w).sum().backward()
        return [xin.grad.clone()] + [None if q.grad is None else q.grad.clone() for q in params]

    # fused
    xf = x.clone().requires_grad_(True)
    y1 = fused.linear_skip(a, lo, xf, cdt, p=p, seed=seed)
    y2 = fused.feed_forward_skip(y1, norm, l1, l2, cdt, p_mid=p, p_out=p, seed=seed + 10)
    gf = grads(y2, xf)
    # reference with the same masks
    m_out = _mask((M, D), p, seed).view(2, M // 2, D)
    m_mid = _mask((M, Fh), p, seed + 10).view(2, M // 2, Fh)
    m_ff = _mask((M, D), p, seed + 11).view(2, M // 2, D)
    xr = x.clone().requires_grad_(True)
    r1 = xr + lo(a.float()) * m_out
    r2 = r1 + l2(F.gelu(l1(norm(r1))) * m_mid) * m_ff
    gr = grads(r2, xr)
    tol = 3e-2 if autocast else 2e-4
    assert C.err_stats(y2.float().cpu(), r2.detach().cpu())["rel_rms"] < tol
    for g1, g0 in zip(gf, gr):
        assert (g1 is None) == (g0 is None)
        if g0 is not None:
            assert C.err_stats(g1.float().cpu(), g0.cpu())["rel_rms"] < 2 * tol


def test_transformer_training_with_dropout_takes_fused_path():
    """dropout: 0.01 as in both reference configs: training mode keeps the fused blocks (mask kernels run), eval mode has
    no dropout, and torch.manual_seed makes the training forward reproducible."""
    torch.manual_seed(0)
    f_dims = {"triv": 0, "se3": 32, "so3": 0, "so2": 32}
    ak = {"f_dims": f_dims, "so2": 8, "so3": 0, "max_freq_h": 1, "max_freq_w": 1}
    tr = gta_amd.Transformer(128, 2, 2, 64, 256, 0.01, True, None, False, {"method": {"name": "gta", "args": ak}}).to(DEV)
    _, ex, x, z = _transformer(False)
    calls = []
    orig = nb.dropout_add
    nb.dropout_add = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        tr.train()
        torch.manual_seed(5)
        y_a = tr(x, None, ex)
        torch.manual_seed(5)
        y_b = tr(x, None, ex)
        torch.manual_seed(6)
        y_c = tr(x, None, ex)
        assert len(calls) == 3 * 4                                   # two per layer
        y_a.sum().backward()
        assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in tr.parameters())
        tr.eval()
        with torch.no_grad():
            y_e = tr(x, None, ex)
        assert len(calls) == 12
    finally:
        nb.dropout_add = orig
    assert torch.equal(y_a, y_b) and not torch.equal(y_a, y_c)
    assert C.err_stats(y_a.detach().cpu(), y_e.cpu())["rel_rms"] < 0.3      # p = 0.01 perturbs, it does not change the answer
