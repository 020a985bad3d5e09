"""Encoder / decoder wrappers around the GTA Transformer -- the ``gta`` path of the reference's
``ImprovedSRTEncoder`` (source/encoder.py:37-345), ``RayPredictor`` + ``ImprovedSRTDecoder``
(source/decoder.py:27-384) and ``TransformingSRT`` (source/models_nvs.py:14-91), so that a whole
forward / backward of the novel-view-synthesis model runs on the HIP attention path.

Same constructor arguments, forward signatures, ``extras`` contract and state-dict keys for the
configurations the GTA runs use (``runs/*/GTA/*/config.yaml``): encoder ``emb: False`` (images only),
decoder ``emb: const`` (one learned query vector).  The other embeddings (ray / planar /
camera_planar) and the competing methods (repast, ape, mln, gbt, frustum_posemb) are the
reference's baselines and stay out of scope; asking for them raises.  The conv stem, the 1x1
projection, LayerNorm, the MLPs and the render MLP are ordinary PyTorch-ROCm modules (MIOpen /
rocBLAS); the rep builders and the attention core are this library's HIP kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.nn import init

from .layers import Transformer
from .reps import pre_compute_reps_decoder, pre_compute_reps_encoder


class SRTLinear(nn.Linear):
    """Xavier-uniform weights, zero bias (layers.py:40-49)."""

    def reset_parameters(self):
        init.xavier_uniform_(self.weight)
        if self.bias is not None:
            init.zeros_(self.bias)


class SRTConvBlock(nn.Module):
    """Two bias-free 3x3 convolutions with ReLU; the second one halves the resolution (encoder.py:13-34)."""

    def __init__(self, idim, hdim=None, odim=None, downsample=True):
        super().__init__()
        hdim = idim if hdim is None else hdim
        odim = 2 * hdim if odim is None else odim
        self.layers = nn.Sequential(
            nn.Conv2d(idim, hdim, kernel_size=3, stride=1, padding=1, bias=False), nn.ReLU(),
            nn.Conv2d(hdim, odim, kernel_size=3, stride=2 if downsample else 1, padding=1, bias=False), nn.ReLU())

    def forward(self, x):
        return self.layers(x)


def _gta_args(attn_args, who):
    method = attn_args["method"]["name"]
    if method != "gta":
        raise NotImplementedError(f"{who}: only attn_args.method.name == 'gta' is built (got {method!r})")
    return attn_args["method"]["args"]


class ImprovedSRTEncoder(nn.Module):
    """images [B,N,3,H,W] -> scene tokens [B, N*h*w, attdim]  (encoder.py:37-345, gta path)."""

    def __init__(self, dim=768, attdim=768, num_conv_blocks=3, num_att_blocks=5, pos_start_octave=0, heads=12,
                 dim_out=None, dropout=None, output_scaler=False, patch_method="conv", emb="ray", attn_args=None,
                 **kwargs):
        super().__init__()
        self.attn_args = _gta_args(attn_args, "ImprovedSRTEncoder")
        if emb is not False:
            raise NotImplementedError("ImprovedSRTEncoder: the GTA configs use emb: False; ray/planar embeddings are not built")
        self.method, self.is_gta, self.emb, self.heads = "gta", True, emb, heads
        blocks = [SRTConvBlock(idim=3, hdim=dim // 8)]
        cur = dim // 4
        for _ in range(1, num_conv_blocks):
            blocks.append(SRTConvBlock(idim=cur, odim=None))
            cur *= 2
        self.conv_blocks = nn.Sequential(*blocks)
        self.per_patch_linear = nn.Conv2d(cur, attdim, kernel_size=1)
        self.transformer = Transformer(dim=attdim, depth=num_att_blocks, heads=heads, dim_head=attdim // heads,
                                       mlp_dim=attdim * 2, selfatt=True, dropout=dropout, attn_args=attn_args)
        self.lin_out = nn.Linear(attdim, dim_out) if dim_out is not None else nn.Identity()
        self.output_scaler = output_scaler
        if output_scaler:
            self.scaler = nn.Parameter(torch.Tensor((output_scaler,)))

    def forward(self, images, camera_pos, rays, extras=None):
        extras = {} if extras is None else extras
        B, N = images.shape[:2]
        pre_compute_reps_encoder(self.attn_args, extras)            # HIP: inverse, Wigner-D, SO(2) table
        x = self.per_patch_linear(self.conv_blocks(images.flatten(0, 1)))
        x = x.flatten(2, 3).permute(0, 2, 1)                        # [B*N, h*w, attdim]
        x = x.reshape(B, N * x.shape[1], x.shape[2])                # view-major tokens
        x = self.lin_out(self.transformer(x, None, extras))
        if self.output_scaler:
            extras["scaler"] = self.scaler
        return x, extras


class RayPredictor(nn.Module):
    """Query tokens -> cross-attention over the scene tokens (decoder.py:27-137, ``emb: const``)."""

    def __init__(self, dim=180, num_att_blocks=2, pos_start_octave=0, z_dim=768, input_mlp=False, heads=12,
                 dim_head=None, mlp_dim=None, return_last_attmap=False, dropout=None, emb="ray", prenorm=True,
                 H=128, W=128, attn_args=None, **kwargs):
        super().__init__()
        if emb != "const":
            raise NotImplementedError("RayPredictor: the GTA configs use emb: const; ray/planar query embeddings are not built")
        self.emb, self.dim, self.return_last_attmap = emb, dim, return_last_attmap
        self.initial_emb = nn.Parameter(torch.randn(dim))
        self.transformer = Transformer(dim, depth=num_att_blocks, heads=heads, dim_head=dim_head, mlp_dim=mlp_dim,
                                       selfatt=False, kv_dim=z_dim, return_last_attmap=return_last_attmap,
                                       dropout=dropout, attn_args=attn_args)

    def forward(self, z, x, rays, extras, queries=None):
        if queries is None:
            queries = self.initial_emb[None, None].expand(rays.shape[0], rays.shape[1], -1)
        return self.transformer(queries, z, extras), z, queries


class ImprovedSRTDecoder(nn.Module):
    """scene tokens + target rays -> pixels [B, T, 3]  (decoder.py:140-384, gta path)."""

    def __init__(self, dim=180, num_att_blocks=2, pos_start_octave=0, z_dim=768, heads=12, return_last_attmap=False,
                 rmlp_dim=1536, act="lrelu", dim_in=None, dropout=None, dim_head=None, mlp_dim=None, emb="ray",
                 prenorm=True, sigmoid=True, attn_args=None, **kwargs):
        super().__init__()
        self.attn_args = _gta_args(attn_args, "ImprovedSRTDecoder")
        self.method, self.is_gta, self.heads = "gta", True, heads
        self.lin_in = nn.Linear(dim_in, z_dim) if dim_in is not None else nn.Identity()
        dim_head = z_dim // heads if dim_head is None else dim_head
        mlp_dim = z_dim * 2 if mlp_dim is None else mlp_dim
        self.allocation_transformer = RayPredictor(
            dim=dim, num_att_blocks=num_att_blocks, pos_start_octave=pos_start_octave, z_dim=z_dim, input_mlp=True,
            heads=heads, dim_head=dim_head, mlp_dim=mlp_dim, return_last_attmap=return_last_attmap, dropout=dropout,
            emb=emb, prenorm=prenorm, attn_args=attn_args, **kwargs)
        assert (not return_last_attmap) or heads == 1
        self.return_last_attmap = return_last_attmap
        acts = {"relu": nn.ReLU, "lrelu": nn.LeakyReLU, "gelu": nn.GELU}
        if act not in acts:
            raise NotImplementedError(act)
        mlp = [SRTLinear(dim, rmlp_dim), acts[act]()]
        for _ in range(3):
            mlp += [SRTLinear(rmlp_dim, rmlp_dim), acts[act]()]
        mlp += [SRTLinear(rmlp_dim, 3), nn.Sigmoid() if sigmoid else nn.Identity()]
        self.render_mlp = nn.Sequential(*mlp)

    def forward(self, z, x, rays, extras):
        z = self.lin_in(z)
        pre_compute_reps_decoder(self.attn_args, extras)             # q side only; keeps the encoder's k side
        out, _, _ = self.allocation_transformer(z, x, rays, extras)
        ret = {}
        if self.return_last_attmap:
            out, attn = out
            ret["masks"] = attn.squeeze(1)
        return self.render_mlp(out), ret


class TransformingSRT(nn.Module):
    """``cfg = {'encoder': 'isrt', 'decoder': 'isrt', 'encoder_kwargs': {...}, 'decoder_kwargs': {...}}``
    (models_nvs.py:14-91).  The ``ftl`` feature-transform variant is not built."""

    def __init__(self, cfg):
        super().__init__()
        if cfg.get("encoder") != "isrt" or cfg.get("decoder") != "isrt":
            raise ValueError("Unknown encoder / decoder type", cfg.get("encoder"), cfg.get("decoder"))
        if cfg.get("ftl", False):
            raise NotImplementedError("TransformingSRT(ftl=True) is not built")
        self.ftl = False
        self.encoder = ImprovedSRTEncoder(**cfg["encoder_kwargs"])
        self.decoder = ImprovedSRTDecoder(**cfg["decoder_kwargs"])

    def decode(self, z, x, rays, extras=None):
        extras = {} if extras is None else extras
        if x.dim() == 4:
            x, rays = x.flatten(1, 2), rays.flatten(1, 2)
        return self.decoder(z, x, rays, extras)

    def forward(self, input_images, input_camera_pos, input_rays, target_camera_pos, target_rays, extras=None):
        extras = {} if extras is None else extras
        z, extras = self.encoder(input_images, input_camera_pos, input_rays, extras)
        return self.decode(z, target_camera_pos, target_rays, extras=extras)


@torch.no_grad()
def render_image(model: "TransformingSRT", z, camera_pos, rays, extras, max_num_rays: int = 8192, reuse_kv: bool = True):
    """Full-image decode of one target view per scene, in query chunks (trainer.py:137-181).

    z [B,K,C] scene tokens from ``model.encoder`` (whose call also left the key-side reps in ``extras``);
    camera_pos [B,3]; rays [B,h,w,3]; ``extras['target_transforms']`` [B,1,4,4] is the pose of the rendered view.
    Returns ``(img [B,h,w,3], {})``.  The reference re-projects K/V and re-applies rho_k for every chunk of every
    layer; here (``reuse_kv``) each cross-attention layer keeps its K/V projection and its K'/V' tile images across
    the chunks (GTA_FLAG_KV_READY), so a chunk costs the query side plus the attention kernel only."""
    from .gta import make_2dcoord
    B, h, w = rays.shape[:3]
    coord = torch.from_numpy(make_2dcoord(h, w)).to(z.device).flatten(0, 1)[None].expand(B, -1, -1)   # [B,h*w,2]
    rays = rays.flatten(1, 2)
    cam = camera_pos[:, None].expand(-1, rays.shape[1], -1)
    img = torch.zeros(B, h * w, 3, dtype=camera_pos.dtype, device=camera_pos.device)
    ex = dict(extras)                                  # the caller's dict keeps its own target_* entries
    if reuse_kv:
        ex["gta_kv_cache"] = {}
    for i in range(0, h * w, max_num_rays):
        sl = slice(i, i + max_num_rays)
        ex["target_rays"] = rays[:, None, sl]
        ex["target_coord"] = coord[:, None, sl]
        pix, _ = model.decode(z=z, x=cam[:, None, sl], rays=rays[:, None, sl], extras=ex)
        img[:, sl] = pix.to(img.dtype)
    return img.view(B, h, w, 3), {}


def mse2psnr(mse: torch.Tensor) -> torch.Tensor:
    """common.py:14-15."""
    return -10.0 * torch.log(mse) / math.log(10.0)


def compute_loss(model, data: dict, mixed_prec: bool = True):
    """The training loss of trainer.py:85-134 for batches that carry ``target_transforms``:
    per-sample MSE over (target pixels, rgb) in fp32; returns (loss [B], {'mse', 'psnr'})."""
    extras = {k: data[k] for k in ("input_transforms", "target_transforms", "input_coord", "target_coord")}
    extras["input_rays"], extras["target_rays"] = data["input_rays"], data["target_rays"]
    target = data["target_pixels"].flatten(1, 2)
    dev = target.device.type
    with torch.autocast(device_type=dev, dtype=torch.bfloat16 if mixed_prec else torch.float32, enabled=mixed_prec):
        pred, _ = model(data["input_images"], data["input_camera_pos"], data["input_rays"],
                        data["target_camera_pos"], data["target_rays"], extras)
    pred = pred.reshape(target.shape).float()
    loss = ((pred - target) ** 2).mean((1, 2))
    return loss, {"mse": loss, "psnr": mse2psnr(loss)}


def synthetic_batch(B, n_in=5, n_tgt=5, image=128, points_per_view=512, device="cuda", seed=0, dtype=torch.float32):
    """A batch with the data loaders' output contract (multishapenet.py / clevr_tr.py): random images,
    canonical first view, patch-centre input coords (three stride-2 stages: stride 8), uniformly sampled target
    coords, random target pixels.  For throughput runs and smoke tests -- there is no dataset in the image."""
    from .gta import make_2dcoord
    g = torch.Generator().manual_seed(seed)

    def poses(n):
        A = torch.randn(B, n, 3, 3, generator=g, dtype=torch.float64)
        Q, R = torch.linalg.qr(A)
        Q = Q * torch.sign(torch.diagonal(R, dim1=-2, dim2=-1))[..., None, :]
        Q[..., :, 0] = Q[..., :, 0] * torch.linalg.det(Q)[..., None]
        E = torch.zeros(B, n, 4, 4, dtype=torch.float64)
        E[..., :3, :3], E[..., :3, 3], E[..., 3, 3] = Q, torch.randn(B, n, 3, generator=g, dtype=torch.float64), 1.0
        E[:, 0] = torch.eye(4, dtype=torch.float64)
        return E.to(dtype)

    ih, iw = (image, image) if isinstance(image, int) else image      # (CLEVR-TR: 120 x 160 inputs, clevr_tr.py:255-260)
    grid = torch.from_numpy(make_2dcoord(ih, iw))
    coord_in = grid[4::8, 4::8].reshape(-1, 2)
    flat = grid.reshape(-1, 2)
    idx = torch.randint(0, flat.shape[0], (B, n_tgt, points_per_view), generator=g)
    batch = {
        "input_images": torch.rand(B, n_in, 3, ih, iw, generator=g, dtype=dtype),
        "input_camera_pos": torch.randn(B, n_in, 3, generator=g, dtype=dtype),
        "input_rays": torch.zeros(B, n_in, 1, 1, 3, dtype=dtype),          # unused on the gta path (emb: False)
        "target_camera_pos": torch.randn(B, n_tgt, points_per_view, 3, generator=g, dtype=dtype),
        "target_rays": torch.randn(B, n_tgt, points_per_view, 3, generator=g, dtype=dtype),
        "target_pixels": torch.rand(B, n_tgt, points_per_view, 3, generator=g, dtype=dtype),
        "input_transforms": poses(n_in), "target_transforms": poses(n_tgt),
        "input_coord": coord_in[None, None].expand(B, n_in, -1, 2).contiguous().to(dtype),
        "target_coord": flat[idx].to(dtype),
    }
    return {k: v.to(device) for k, v in batch.items()}


def msn_gta_so3_cfg(dropout=0.01):
    """model.args of runs/msn/GTA/gta_so3/config.yaml."""
    args = dict(so2=6, so3=2, max_freq_h=1, max_freq_w=1, f_dims=dict(triv=0, se3=48, so2=24, so3=24))
    aa = {"method": {"name": "gta", "args": args}}
    return {"encoder": "isrt", "decoder": "isrt",
            "encoder_kwargs": dict(pos_start_octave=-5, dropout=dropout, heads=8, emb=False, attn_args=aa),
            "decoder_kwargs": dict(z_dim=768, pos_start_octave=-5, dropout=dropout, heads=8, emb="const",
                                   attn_args={"method": {"name": "gta", "args": dict(args)}})}


def clevrtr_gta_cfg(dropout=0.01, precise=False):
    """model.args of runs/clevrtr/GTA/gta/config.yaml:14-52 (SE(3) + SO(2) reps; 2 input views of 120 x 160 -> 600 scene tokens, dh = 64).
    ``precise`` = this build's fp32-faithful arithmetic for the config's ``mixed_prec: False`` (config.yaml:55)."""
    def aa():
        args = dict(so2=8, max_freq_h=1, max_freq_w=1, f_dims=dict(se3=32, so2=32))
        if precise:
            args["precise"] = True
        return {"method": {"name": "gta", "args": args}}
    return {"encoder": "isrt", "decoder": "isrt",
            "encoder_kwargs": dict(pos_start_octave=-5, dim=768, attdim=384, heads=6, dropout=dropout, emb=False, attn_args=aa()),
            "decoder_kwargs": dict(z_dim=384, rmlp_dim=768, heads=6, pos_start_octave=-5, dropout=dropout, emb="const", attn_args=aa())}
