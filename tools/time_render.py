"""Full-image decode timing: MSN gta_so3 model, one 128x128 target view per scene,
chunked queries, with and without the per-layer K/V cache.  Usage: python tools/time_render.py [B] [chunk]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from gta_amd import srt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
modes = {"both": (False, True, False, True), "reuse": (True,), "plain": (False,)}[sys.argv[3] if len(sys.argv) > 3 else "both"]
torch.manual_seed(0)
model = srt.TransformingSRT(srt.msn_gta_so3_cfg(dropout=0.0)).cuda().eval()
data = srt.synthetic_batch(B, n_in=5, n_tgt=1, image=128, points_per_view=512, device="cuda", seed=1)
extras = {"input_transforms": data["input_transforms"], "input_coord": data["input_coord"],
          "target_transforms": data["target_transforms"][:, :1]}
rays = torch.randn(B, 128, 128, 3, device="cuda")
cam = torch.randn(B, 3, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    z, extras = model.encoder(data["input_images"], data["input_camera_pos"], data["input_rays"], extras)
    for reuse in modes:
        for _ in range(6):
            img, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=chunk, reuse_kv=reuse)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            img, _ = srt.render_image(model, z, cam, rays, extras, max_num_rays=chunk, reuse_kv=reuse)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B} 128x128 view, chunk={chunk}, reuse_kv={reuse}: {dt * 1e3:.2f} ms / image batch, "
              f"{B * 128 * 128 / dt / 1e6:.2f} Mpixel/s, finite={bool(torch.isfinite(img).all())}")
