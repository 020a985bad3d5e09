"""2-D GTA for single-image transformers (the reference's DiT branch, README.md:24,32,36).

That branch is not part of the checkout; what it uses of GTA is the pure-SO(2) case of the operator -- every token of an
h x w patch grid carries rotations by its (row, column) coordinate, no camera poses (in-tree analogue: the encoder of
``runs/clevrtr/GTA/gta_no3demb/config.yaml``, ``f_dims: {so2: 64}``, ``so2: 16``).  ``GTA2DTransformer`` packages that:
a ``gta_amd.Transformer`` whose attention runs the fused kernels on the ``GTA_LAYOUT_SO2`` specialisation, with the
(cos, sin) table of the grid built once per (batch size, device) by ``gta_build_so2_table``.

The wrapped blocks keep the reference's parameter names (``layers.*``), so weights of a reference ``Transformer`` built
with the same ``attn_args`` load with ``strict=True`` into ``.transformer``."""
from __future__ import annotations

import torch
from torch import nn

from . import native
from .gta import make_2dcoord
from .layers import Transformer


class GTA2DTransformer(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int, grid, dropout: float = 0.0,
                 max_freq_h: float = 1.0, max_freq_w: float = 1.0, shared_freqs: bool = False):
        super().__init__()
        if dim_head % 4:
            raise ValueError("pure-SO(2) GTA needs dim_head % 4 == 0 (2x2 blocks, two coordinates per frequency)")
        self.grid = (int(grid[0]), int(grid[1]))
        self.attn_kwargs = {"f_dims": {"so2": dim_head}, "so2": dim_head // 4, "so3": 0, "max_freq_h": max_freq_h,
                            "max_freq_w": max_freq_w, "shared_freqs": shared_freqs}
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout, True, None, False,
                                       {"method": {"name": "gta", "args": self.attn_kwargs}})
        self._tables = {}

    def reps(self, B: int, device) -> dict:
        """extras with the packed (cos, sin) table of the patch grid for B images (q side == k side)."""
        key = (B, str(device))
        if key not in self._tables:
            h, w = self.grid
            coord = torch.from_numpy(make_2dcoord(h, w)).reshape(1, h * w, 2).to(device).expand(B, -1, -1).contiguous()
            ak = self.attn_kwargs
            cs = native.build_so2_table(coord, ak["so2"], ak["max_freq_h"], ak["max_freq_w"], ak["shared_freqs"])
            self._tables = {key: cs}                      # one entry: a new batch size replaces the old table
        cs = self._tables[key]
        return {"gta_cs_q": cs, "gta_cs_k": cs}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, h*w, dim] -> [B, h*w, dim]"""
        h, w = self.grid
        if x.shape[1] != h * w:
            raise ValueError(f"expected {h * w} tokens (a {h}x{w} grid), got {x.shape[1]}")
        return self.transformer(x, None, self.reps(x.shape[0], x.device))
