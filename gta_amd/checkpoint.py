"""Reader / writer for the reference's checkpoint files (source/checkpoint.py:5-59, written by train.py:301-308).

A checkpoint is one ``torch.save``d dict: a state dict per registered module -- the reference registers
``encoder``, ``decoder`` and ``optimizer`` -- plus scalars (``epoch_it``, ``it``, ``t``, ``loss_val_best``,
``run_id``).  ``gta_amd.TransformingSRT`` keeps the reference's parameter names, so the ``encoder`` / ``decoder``
entries load with ``strict=True``.  DDP-wrapped modules saved with a ``module.`` prefix are accepted too.
"""
from __future__ import annotations

import os
from typing import Dict

import torch


def _strip_module(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_checkpoint(path: str, device=None, trusted: bool = False, **modules) -> dict:
    """``load_checkpoint(path, encoder=model.encoder, decoder=model.decoder, optimizer=opt)`` loads every
    registered module found in the file (strict) and returns the remaining entries, like ``Checkpoint.load``.

    The file is read with ``weights_only=True`` (state dicts and scalars, which is all the reference's released
    checkpoints hold); ``trusted=True`` allows arbitrary pickles for files you wrote yourself."""
    state = torch.load(path, map_location=device, weights_only=not trusted)
    for name, mod in modules.items():
        if name not in state:
            raise KeyError(f"checkpoint {os.path.basename(path)} has no entry '{name}' (found: {sorted(state)})")
        sd = state[name]
        if isinstance(mod, torch.nn.Module):
            mod.load_state_dict(_strip_module(sd), strict=True)
        else:
            mod.load_state_dict(sd)
    return {k: v for k, v in state.items() if k not in modules}


def save_checkpoint(path: str, extra: dict | None = None, **modules) -> None:
    """Writes the reference layout: one state dict per module plus the extra scalars."""
    out = dict(extra or {})
    for name, mod in modules.items():
        m = mod.module if hasattr(mod, "module") and isinstance(mod, torch.nn.parallel.DistributedDataParallel) else mod
        out[name] = m.state_dict()
    torch.save(out, path)
