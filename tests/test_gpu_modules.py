"""GPU: the drop-in Transformer/Attention modules (gta_amd.layers) against the reference module
fixtures (tests/golden/mod_*.npz: reference weights, inputs, outputs, input- and parameter-grads)
and PSNR parity of a rendered-pixel proxy (common.py:14-15)."""
import math

import pytest
import torch

import gta_amd
from tests import _golden as G
from tests import _hip_cases as C

pytestmark = pytest.mark.gpu


def _build(case, dtype=torch.float32):
    d, meta = G.load("mod_" + case)
    ak = {"f_dims": meta["f_dims"], "so2": meta["so2"], "so3": meta["so3"], "max_freq_h": 1, "max_freq_w": 1}
    tr = gta_amd.Transformer(meta["dim"], meta["depth"], meta["H"], meta["dh"], 2 * meta["dim"], 0.0,
                             not meta["cross"], meta["kv_dim"], False, {"method": {"name": "gta", "args": ak}})
    sd = {k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")}
    tr.load_state_dict(sd, strict=True)          # reference checkpoint keys load unchanged
    tr = tr.cuda()
    ex = {k[len("extras."):]: torch.from_numpy(v).float().cuda() for k, v in d.items()
          if k in ("extras.input_transforms", "extras.target_transforms", "extras.input_coord", "extras.target_coord")}
    gta_amd.pre_compute_reps_encoder(ak, ex)
    if meta["cross"]:
        gta_amd.pre_compute_reps_decoder(ak, ex)
    return d, meta, tr, ex


@pytest.mark.parametrize("case", G.list_cases("mod_"))
def test_transformer_forward_backward_vs_reference(case):
    d, meta, tr, ex = _build(case)
    x = torch.from_numpy(d["x"]).float().cuda().requires_grad_()
    z = torch.from_numpy(d["z"]).float().cuda() if "z" in d else None
    y = tr(x, z, ex)
    (y * torch.from_numpy(d["w"]).float().cuda()).sum().backward()
    torch.cuda.synchronize()
    st = C.err_stats(y.detach().cpu(), torch.from_numpy(d["y"]).float())
    assert st["finite"] and st["rel_rms"] < 1e-2 and st["max_abs"] < 3e-2 * st["ref_max"], st
    st = C.err_stats(x.grad.cpu(), torch.from_numpy(d["dx"]).float())
    assert st["finite"] and st["rel_rms"] < 3e-2, st
    for n, p in tr.named_parameters():
        ref = torch.from_numpy(d["grad." + n]).float()
        st = C.err_stats(p.grad.cpu(), ref)
        assert st["finite"], (n, st)
        if n.endswith("trans_coeff"):
            # one scalar summed over every token and se3 block with heavy cancellation (|true| ~ 0.1 here):
            # the bf16-product noise of that sum is ~0.05 absolute, the level the operator tests
            # (test_gpu_backward.py, |true| 1..20, error <= 0.6 %) also show.  Absolute bound.
            assert st["max_abs"] <= 0.12, (n, st)
            continue
        assert st["max_abs"] <= 4e-2 * max(st["ref_max"], 1e-3) + 1e-5, (n, st)


def test_psnr_parity_of_module_outputs():
    """'PSNR parity' (BASELINE.md): |PSNR_hip - PSNR_ref| of a sigmoid-rendered proxy of the module
    output against a fixed target, under identical weights -- mse2psnr as common.py:14-15."""
    d, meta, tr, ex = _build("dec_ms")
    x = torch.from_numpy(d["x"]).float().cuda()
    z = torch.from_numpy(d["z"]).float().cuda()
    y_hip = tr(x, z, ex).detach().cpu()
    y_ref = torch.from_numpy(d["y"]).float()
    g = torch.Generator().manual_seed(0)
    target = torch.rand(y_ref.shape, generator=g)
    psnr = lambda y: -10.0 * math.log10(((torch.sigmoid(y) - target) ** 2).mean().item())
    assert abs(psnr(y_hip) - psnr(y_ref)) < 0.02


def test_autocast_bf16_module():
    """Under torch.autocast(bf16) the projections emit bf16 and the kernel runs its bf16-input path."""
    d, meta, tr, ex = _build("enc_cl")
    x = torch.from_numpy(d["x"]).float().cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = tr(x, None, ex)
    st = C.err_stats(y.float().cpu(), torch.from_numpy(d["y"]).float())
    assert st["finite"] and st["rel_rms"] < 3e-2, st


def _srt(dtype=torch.float32, fixture="srt_ms_tiny"):
    import ast
    import numpy as np
    from gta_amd import srt
    d, _ = G.load(fixture)
    cfg = ast.literal_eval(str(np.load(G.GOLDEN + f"/{fixture}.npz")["meta"]))
    model = srt.TransformingSRT(cfg)
    model.load_state_dict({k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")},
                          strict=True)
    t = lambda n: torch.from_numpy(d[n]).float().cuda()
    data = {"input_images": t("images"), "input_camera_pos": t("cam_in"), "input_rays": t("rays_in"),
            "target_camera_pos": t("cam_t"), "target_rays": t("rays_t"), "target_pixels": t("target"),
            "input_transforms": t("extras.input_transforms"), "target_transforms": t("extras.target_transforms"),
            "input_coord": t("extras.input_coord"), "target_coord": t("extras.target_coord")}
    return d, model.cuda(), data


def _srt_grad_check(model, d, mixed):
    """parameter gradients of the model against the reference's (fixture); returns the worst relative RMS"""
    worst = 0.0
    for n, p in model.named_parameters():
        ref = torch.from_numpy(d["grad." + n]).float()
        st = C.err_stats(p.grad.cpu(), ref)
        assert st["finite"], (n, st)
        if n.endswith("trans_coeff"):
            continue                                  # cancellation-dominated scalar, checked at operator level
        tol = (2e-1 if mixed else 5e-2) * max(st["ref_max"], 1e-4) + 1e-6
        assert st["max_abs"] <= tol, (n, st)
        worst = max(worst, st["rel_rms"])
    assert worst < (0.4 if mixed else 0.1), worst
    return worst


@pytest.mark.parametrize("fixture,mixed", [("srt_ms_tiny", False), ("srt_ms_rays", False), ("srt_ms_rays", True)])
def test_srt_model_matches_reference(fixture, mixed):
    """Whole TransformingSRT forward + loss + backward on the HIP path under the reference's weights: rendered pixels,
    per-sample MSE, PSNR parity and every parameter gradient.  fp32 on both fixtures (2 x 5 and 2 x 128 rays per scene);
    bf16 autocast on the 256-ray one with FOUR TIMES the fp32 bounds.  (On the 10-ray fixture a bf16 run has no meaningful
    gradient bound: the render MLP's LeakyReLU units whose near-zero pre-activation changes sign against the fp32 reference
    change their gradient by 1 / slope = 100x, and with 10 rays one flip moves every gradient upstream by tens of percent --
    profiles/r02/README.md, "SRT fixture under bf16"; with 256 rays the flips average out.)"""
    from gta_amd import srt
    d, model, data = _srt(fixture=fixture)
    loss, terms = srt.compute_loss(model, data, mixed_prec=mixed)
    loss.sum().backward()
    torch.cuda.synchronize()
    ref_loss = torch.from_numpy(d["loss"]).float()
    ref_psnr = torch.from_numpy(d["psnr"]).float()
    assert (loss.cpu() - ref_loss).abs().max() <= (4e-2 if mixed else 1e-2) * ref_loss.abs().max()
    assert (terms["psnr"].detach().cpu() - ref_psnr).abs().max() <= (0.2 if mixed else 0.05)       # dB
    _srt_grad_check(model, d, mixed)


@pytest.mark.parametrize("fname", ["ckpt_ref_ms.pt", "ckpt_ref_ms_ddp.pt"])
def test_checkpoint_written_by_the_reference_renders_on_the_hip_path(fname):
    """a checkpoint file written by the reference's own ``Checkpoint.save`` (checkpoint.py:21-35; fixture from oracle/make_golden.py
    ``checkpoint_case``: fp32 weights after one AdamW step, encoder / decoder / optimizer + scalars; the ``_ddp`` file with the ``module.`` prefix)
    -> ``gta_amd.checkpoint.load_checkpoint`` -> HIP forward of the whole TransformingSRT -> the pixels the REFERENCE model rendered from that file.
    (The J-convention half of f3 needs the released J_dense.pt, which is not in the image: parked)"""
    import ast
    import numpy as np
    from gta_amd import srt, checkpoint
    io = np.load(G.GOLDEN + "/ckpt_ref_ms_io.npz")
    cfg = ast.literal_eval(str(io["meta"]))
    model = srt.TransformingSRT(cfg)
    rest = checkpoint.load_checkpoint(G.GOLDEN + "/" + fname, device="cuda", encoder=model.encoder, decoder=model.decoder)
    assert rest["it"] == 1234 and rest["run_id"] == "golden"
    model = model.cuda().eval()
    t = lambda n: torch.from_numpy(io[n]).float().cuda()
    ex = {k[len("extras."):]: torch.from_numpy(io[k]).float().cuda() for k in io.files if k.startswith("extras.")}
    ref = torch.from_numpy(io["pred"]).float()
    B, NtP = ref.shape[0], ref.shape[1]
    from gta_amd import layers
    for precise in (False, True):             # default (bf16 products) and fp32-faithful arithmetic
        for m in model.modules():
            if isinstance(m, layers.Attention):
                m.precise = precise
        with torch.no_grad():
            pred, _ = model(t("images"), t("cam_in"), t("rays_in"), t("cam_t"), t("rays_t"), dict(ex))
        torch.cuda.synchronize()
        st = C.err_stats(pred.reshape(B, NtP, 3).cpu(), ref)
        assert st["finite"] and st["max_abs"] <= (1e-4 if precise else 1e-2) and st["rel_rms"] <= (2e-5 if precise else 5e-3), (precise, st)
        psnr = lambda x: -10.0 * torch.log10(((x - torch.from_numpy(io["target"]).float().flatten(1, 2)) ** 2).mean((1, 2)))
        assert (psnr(pred.reshape(B, NtP, 3).cpu()) - psnr(ref)).abs().max() <= 0.05        # dB


def test_srt_clevr_layout_fp32_faithful_psnr_parity():
    """The CLEVR-TR layout (se3 + so2, no so3: runs/clevrtr/GTA/gta/config.yaml:19-52), the config the reference trains in fp32
    (`mixed_prec: False`, config.yaml:55): whole TransformingSRT under the reference's weights (fixture srt_cl_rays: 2 x 128 rays per
    scene, generated by oracle/make_golden.py from the imported reference) with every attention module in the fp32-faithful mode
    (`Attention.precise`: fp32 rho, split-bf16 forward products, exact-fp32 backward).  PSNR parity to 0.01 dB, loss to 1e-4, every
    parameter gradient an order of magnitude inside the default mode's bounds -- and the default (bf16-product) mode measurably
    further away on the same fixture."""
    from gta_amd import layers, srt
    res = {}
    for mode in ("precise", "default"):
        d, model, data = _srt(fixture="srt_cl_rays")
        for m in model.modules():
            if isinstance(m, layers.Attention):
                m.precise = mode == "precise"
        loss, terms = srt.compute_loss(model, data, mixed_prec=False)
        loss.sum().backward()
        torch.cuda.synchronize()
        ref_loss = torch.from_numpy(d["loss"]).float()
        ref_psnr = torch.from_numpy(d["psnr"]).float()
        worst = 0.0
        for n, p in model.named_parameters():
            if n.endswith("trans_coeff"):
                continue
            st = C.err_stats(p.grad.cpu(), torch.from_numpy(d["grad." + n]).float())
            assert st["finite"], (n, st)
            worst = max(worst, st["max_abs"] / max(st["ref_max"], 1e-4))
        res[mode] = ((loss.cpu() - ref_loss).abs().max().item() / ref_loss.abs().max().item(),
                     (terms["psnr"].detach().cpu() - ref_psnr).abs().max().item(), worst)
    dl, dp, dg = res["precise"]
    assert dl <= 1e-4 and dp <= 0.01 and dg <= 5e-3, res
    assert res["default"][2] > 3 * dg or res["default"][0] > 3 * dl, res       # (the default mode's bf16 products are visible on this fixture)


def test_srt_bf16_gradient_bound_is_not_vacuous():
    """The mixed-precision bounds above reject a wrong decoder gradient: zeroed, doubled or sign-flipped by hand."""
    from gta_amd import srt
    d, model, data = _srt(fixture="srt_ms_rays")
    loss, _ = srt.compute_loss(model, data, mixed_prec=True)
    loss.sum().backward()
    torch.cuda.synchronize()
    _srt_grad_check(model, d, True)
    names = [n for n, p in model.named_parameters() if n.startswith("decoder.") and n.endswith("weight") and p.grad.abs().max() > 0]
    victims = [n for n in names if "transformer" in n][:1] + [n for n in names if "render_mlp" in n][:1]
    assert len(victims) == 2, names
    params = dict(model.named_parameters())
    for n in victims:
        keep = params[n].grad.clone()
        for bad in (torch.zeros_like(keep), 2.0 * keep, -keep):
            params[n].grad = bad
            with pytest.raises(AssertionError):
                _srt_grad_check(model, d, True)
        params[n].grad = keep
    _srt_grad_check(model, d, True)


def test_srt_encoder_blocks_bf16_stream():
    """The SRT encoder's Transformer exactly as the model calls it under mixed precision -- bf16 residual stream (the conv
    stem runs under autocast), d = 48, two heads -- fused blocks against the fp32 module-by-module path: output and all
    gradients for one and the same upstream gradient (no activation masks in between)."""
    from gta_amd import layers, srt
    d, model, data = _srt()
    cap = {}
    tr = model.encoder.transformer
    h = tr.register_forward_pre_hook(lambda m, a: cap.update(x=a[0].detach(), ex=dict(a[2])))
    srt.compute_loss(model, data, mixed_prec=True)
    h.remove()
    assert cap["x"].dtype == torch.bfloat16
    res = {}
    try:
        for name, fused_on, ac in (("ref", False, False), ("fused", True, True)):
            tr.fused_blocks = fused_on
            for p in tr.parameters():
                p.grad = None
            xi = (cap["x"].float() if not ac else cap["x"]).clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
                y = tr(xi, None, cap["ex"])
            w = torch.randn(y.shape, device=y.device, generator=torch.Generator(device=y.device).manual_seed(1))
            (y.float() * w).sum().backward()
            res[name] = (y.detach().float().cpu(), xi.grad.float().cpu(), {n: p.grad.float().cpu() for n, p in tr.named_parameters()})
    finally:
        tr.fused_blocks = True
    assert res["fused"][0].dtype == torch.float32
    assert C.err_stats(res["fused"][0], res["ref"][0])["rel_rms"] < 2e-2
    assert C.err_stats(res["fused"][1], res["ref"][1])["rel_rms"] < 4e-2
    for n, g in res["ref"][2].items():
        if not n.endswith("trans_coeff"):
            assert C.err_stats(res["fused"][2][n], g)["rel_rms"] < 4e-2, n


def test_render_image_chunked_decode():
    """Full-image decode (trainer.py:137-181) under the reference's weights: chunked queries with the per-layer K/V
    cache == the same without the cache (bit for bit) == the oracle decoding every pixel in one call."""
    from gta_amd import srt
    from oracle import gta_oracle as O
    import ast
    import numpy as np
    d, model, data = _srt()
    model.eval()
    h, w = 9, 14
    B = data["input_images"].shape[0]
    g = torch.Generator().manual_seed(3)
    rays = torch.randn(B, h, w, 3, generator=g).cuda()
    cam = torch.randn(B, 3, generator=g).cuda()
    extras = {"input_transforms": data["input_transforms"], "input_coord": data["input_coord"],
              "target_transforms": data["target_transforms"][:, 1:2]}
    with torch.no_grad():
        z, extras = model.encoder(data["input_images"], data["input_camera_pos"], data["input_rays"], extras)
        img_c, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=50, reuse_kv=True)
        img_n, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=50, reuse_kv=False)
        img_1, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=h * w, reuse_kv=True)
    torch.cuda.synchronize()
    assert img_c.shape == (B, h, w, 3)
    # 50-ray chunks: without the cache the layer picks the single-kernel plan (Tq <= 256), with it the two-stage plan --
    # two kernels, the same arithmetic up to fp32 contraction: equal to the last bits, not necessarily bit for bit
    assert (img_c - img_n).abs().max() < 1e-6
    assert (img_c - img_1).abs().max() < 2e-3          # another chunking = other query tiles per workgroup, same pixels
    # same plan on both sides (chunks above 256 rays: two-stage): the cache changes NOTHING, bit for bit
    h2, w2 = 20, 30
    rays2 = torch.randn(B, h2, w2, 3, generator=g).cuda()
    with torch.no_grad():
        a, _ = srt.render_image(model, z, cam, rays2, extras, max_num_rays=300, reuse_kv=True)
        b2, _ = srt.render_image(model, z, cam, rays2, extras, max_num_rays=300, reuse_kv=False)
    assert torch.equal(a, b2)
    # oracle: every pixel of the view as one query set
    cfg = ast.literal_eval(str(np.load(G.GOLDEN + "/srt_ms_tiny.npz")["meta"]))
    om = O.OracleSRT(cfg)
    om.load_state_dict({k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")},
                       strict=True)
    om.eval()
    coord = torch.from_numpy(gta_amd.gta.make_2dcoord(h, w)).flatten(0, 1)[None, None].expand(B, 1, -1, -1)
    ex_o = {"input_transforms": data["input_transforms"].cpu(), "input_coord": data["input_coord"].cpu(),
            "target_transforms": data["target_transforms"][:, 1:2].cpu(), "target_coord": coord}
    with torch.no_grad():
        ref = om(data["input_images"].cpu(), None, None, None, rays.cpu().flatten(1, 2), ex_o).view(B, h, w, 3)
    st = C.err_stats(img_c.cpu(), ref)
    assert st["finite"] and st["max_abs"] < 1e-2, st
    mse = ((img_c.cpu() - ref) ** 2).mean()
    assert mse < 1e-5, mse


def test_render_image_full_size_view():
    """Full-image decode at the evaluation size (trainer.py:137-181, evaluate.py:122-131): one 128 x 128 target view per
    scene = 16 384 query rays in 2 048-ray chunks with the per-layer K/V cache, against the oracle decoding all pixels
    in one call under the reference's weights (fixture srt_ms_tiny).  Every pixel is compared."""
    from gta_amd import srt
    from oracle import gta_oracle as O
    import ast
    import numpy as np
    d, model, data = _srt()
    model.eval()
    h = w = 128
    B = data["input_images"].shape[0]
    g = torch.Generator().manual_seed(5)
    rays = torch.nn.functional.normalize(torch.randn(B, h, w, 3, generator=g), dim=-1).cuda()
    cam = torch.randn(B, 3, generator=g).cuda()
    extras = {"input_transforms": data["input_transforms"], "input_coord": data["input_coord"],
              "target_transforms": data["target_transforms"][:, 1:2]}
    with torch.no_grad():
        z, extras = model.encoder(data["input_images"], data["input_camera_pos"], data["input_rays"], extras)
        img, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=2048, reuse_kv=True)
    torch.cuda.synchronize()
    assert img.shape == (B, h, w, 3)
    cfg = ast.literal_eval(str(np.load(G.GOLDEN + "/srt_ms_tiny.npz")["meta"]))
    om = O.OracleSRT(cfg)
    om.load_state_dict({k[len("param."):]: torch.from_numpy(v).float() for k, v in d.items() if k.startswith("param.")},
                       strict=True)
    om.eval()
    coord = torch.from_numpy(gta_amd.gta.make_2dcoord(h, w)).flatten(0, 1)[None, None].expand(B, 1, -1, -1)
    ex_o = {"input_transforms": data["input_transforms"].cpu(), "input_coord": data["input_coord"].cpu(),
            "target_transforms": data["target_transforms"][:, 1:2].cpu(), "target_coord": coord}
    with torch.no_grad():
        ref = om(data["input_images"].cpu(), None, None, None, rays.cpu().flatten(1, 2), ex_o).view(B, h, w, 3)
    st = C.err_stats(img.cpu(), ref)
    assert st["finite"] and st["max_abs"] < 1e-2, st
    assert ((img.cpu() - ref) ** 2).mean() < 1e-5


def test_render_image_clevrtr_size_view():
    """Full-image decode at the CLEVR-TR evaluation size (trainer.py:137-181, evaluate.py:122-131; SURVEY 8 f4 names both sizes): one
    240 x 320 target view per scene = 76 800 query rays against the 600 scene tokens of two 120 x 160 input views, CLEVR-TR layout (se3 32 + so2 32,
    dh = 64: the gta_fwd2 / KV_READY instances), in 20 480-ray chunks (the reference's max_num_rays = num_points * batch_size / B at B = 4,
    trainer.py:155-157) with the per-layer K'/V' cache reused across the chunks.  The oracle decodes 2 400 sampled pixels per scene (every query of a
    cross-attention decoder is independent of the others) under the same weights; the last chunk is ragged (76 800 = 3 x 20 480 + 15 360)."""
    from gta_amd import srt
    from oracle import gta_oracle as O
    torch.manual_seed(7)
    aa = {"method": {"name": "gta", "args": dict(so2=8, max_freq_h=1, max_freq_w=1, f_dims=dict(se3=32, so2=32))}}
    cfg = {"encoder": "isrt", "decoder": "isrt",
           "encoder_kwargs": dict(pos_start_octave=-5, dim=96, attdim=128, num_att_blocks=2, heads=2, dropout=0.0, emb=False, attn_args=aa),
           "decoder_kwargs": dict(dim=36, z_dim=128, rmlp_dim=64, heads=2, pos_start_octave=-5, dropout=0.0, emb="const", attn_args=aa)}
    model = srt.TransformingSRT(cfg).cuda().eval()
    B, h, w = 2, 240, 320
    data = srt.synthetic_batch(B, n_in=2, n_tgt=1, image=(120, 160), points_per_view=8, device="cuda", seed=3)
    assert data["input_coord"].shape[2] == 300                                     # 15 x 20 patch tokens per view -> Tk = 600
    g = torch.Generator().manual_seed(5)
    rays = torch.nn.functional.normalize(torch.randn(B, h, w, 3, generator=g), dim=-1).cuda()
    cam = torch.randn(B, 3, generator=g).cuda()
    extras = {"input_transforms": data["input_transforms"], "input_coord": data["input_coord"],
              "target_transforms": data["target_transforms"][:, :1]}
    with torch.no_grad():
        z, extras = model.encoder(data["input_images"], data["input_camera_pos"], data["input_rays"], extras)
        assert z.shape == (B, 600, 128)
        img, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=20480, reuse_kv=True)
        img_nc, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=32768, reuse_kv=False)    # other chunking, no cache
    torch.cuda.synchronize()
    assert img.shape == (B, h, w, 3) and bool(torch.isfinite(img).all())
    assert (img - img_nc).abs().max() < 2e-3                                        # cached vs recomputed K'/V', different chunk edges
    om = O.OracleSRT(cfg)
    om.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    om.eval()
    n_s = 2400
    idx = torch.stack([torch.randperm(h * w, generator=g)[:n_s] for _ in range(B)])              # [B, n_s]
    idx[:, :4] = torch.tensor([0, w - 1, (h - 1) * w, h * w - 1])                                 # the four corners ride along
    coord_all = torch.from_numpy(gta_amd.gta.make_2dcoord(h, w)).flatten(0, 1)
    rays_s = torch.gather(rays.cpu().flatten(1, 2), 1, idx[..., None].expand(-1, -1, 3))
    ex_o = {"input_transforms": data["input_transforms"].cpu(), "input_coord": data["input_coord"].cpu(),
            "target_transforms": data["target_transforms"][:, :1].cpu(), "target_coord": coord_all[idx][:, None]}
    with torch.no_grad():
        ref = om(data["input_images"].cpu(), None, None, None, rays_s, ex_o).view(B, n_s, 3)
    got = torch.gather(img.cpu().flatten(1, 2), 1, idx[..., None].expand(-1, -1, 3))
    st = C.err_stats(got, ref)
    assert st["finite"] and st["max_abs"] < 1e-2, st
    assert ((got - ref) ** 2).mean() < 1e-5


def test_gta2d_transformer_dit_shape_vs_oracle():
    """The pure-SO(2) 2-D GTA block at the DiT stress shape (BASELINE config 5: 32 x 32 patch grid = 1024 tokens, 16
    heads x 64 channels, so2 = 16 frequencies) against the oracle's Transformer under the same weights; forward, input
    gradient and parameter gradients.  (The reference fixture of this layout is mod_dit_so2, run by
    test_transformer_forward_backward_vs_reference.)"""
    from oracle import gta_oracle as O
    torch.manual_seed(0)
    dim, depth, H, dh, grid, B = 128, 1, 16, 64, (32, 32), 2
    m = gta_amd.GTA2DTransformer(dim, depth, H, dh, 2 * dim, grid).cuda()
    ak = dict(m.attn_kwargs)
    om = O.OracleTransformer(dim, depth, H, dh, 2 * dim, 0.0, True, None, False, {"method": {"name": "gta", "args": ak}})
    om.load_state_dict({k: v.detach().cpu() for k, v in m.transformer.state_dict().items()}, strict=True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, grid[0] * grid[1], dim, generator=g)
    wgt = torch.randn(B, grid[0] * grid[1], dim, generator=g)
    coord = torch.from_numpy(gta_amd.make_2dcoord(*grid)).reshape(1, 1, -1, 2).expand(B, 1, -1, -1).float()
    reps = O.encoder_reps(ak, {"input_coord": coord, "input_transforms": torch.eye(4).repeat(B, 1, 1, 1)})
    xo = x.clone().requires_grad_()
    yo = om(xo, None, reps)
    (yo * wgt).sum().backward()
    xd = x.cuda().requires_grad_()
    yd = m(xd)
    (yd * wgt.cuda()).sum().backward()
    torch.cuda.synchronize()
    st = C.err_stats(yd.detach().cpu(), yo.detach())
    assert st["finite"] and st["rel_rms"] < 1e-2 and st["max_abs"] < 3e-2 * st["ref_max"], st
    st = C.err_stats(xd.grad.cpu(), xo.grad)
    assert st["finite"] and st["rel_rms"] < 3e-2, st
    go = dict(om.named_parameters())
    for n, p in m.transformer.named_parameters():
        st = C.err_stats(p.grad.cpu(), go[n].grad)
        assert st["finite"] and st["max_abs"] <= 5e-2 * max(st["ref_max"], 1e-3) + 1e-5, (n, st)
    # the module reuses its coordinate table and the kernels ran the SO2 layout (no view records, dh = 64)
    assert m.reps(B, xd.device)["gta_cs_q"].shape == (B, 1024, 32, 2)
