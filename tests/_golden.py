"""Helpers to load the committed reference fixtures (tests/golden/*.npz)."""
import ast
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def list_cases(prefix):
    return sorted(os.path.basename(p)[len(prefix):-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = ast.literal_eval(str(d.pop("meta"))) if "meta" in d else {}
    return d, meta


def extras_of(d, dtype=torch.float64, device="cpu"):
    """Rebuild the reference's `extras` dict (tensors and per-degree lists) from a fixture."""
    ex, lists = {}, {}
    for k, v in d.items():
        if not k.startswith("extras."):
            continue
        key = k[len("extras."):]
        t = torch.from_numpy(v).to(dtype).to(device)
        if "." in key:
            base, idx = key.rsplit(".", 1)
            lists.setdefault(base, {})[int(idx)] = t
        else:
            ex[key] = t
    for base, items in lists.items():
        ex[base] = [items[i] for i in sorted(items)]
    return ex


def attn_kwargs_of(meta):
    ak = {"f_dims": dict(meta["f_dims"]), "so2": meta["so2"], "so3": meta["so3"],
          "max_freq_h": 1, "max_freq_w": 1}
    ak.update(meta.get("extra", {}))
    return ak


def tau_of(d, dtype=torch.float32, device="cpu", grad=True):
    """The softmax temperature of a `softmax: adjustable` fixture as a 1-element leaf tensor, else None."""
    if "tau" not in d or float(d["tau"]) == 1.0:
        return None
    return torch.tensor([float(d["tau"])], dtype=dtype, device=device, requires_grad=grad)
