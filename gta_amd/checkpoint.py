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


# ---------------------------------------------------------------------------------------------------------------------
# J-convention check
# ---------------------------------------------------------------------------------------------------------------------
# The reference reads the Pinchon-Hoggan J matrices from ``J_dense.pt`` (wigner_d.py:8-9; a third-party blob that is not
# in the checkout), this build carries a restatement (gta_reps.hip kJ1 / kJ2).  A released so3 checkpoint was trained
# with the blob's J, so before serving it the two must be compared.  The only freedom a correct J has is the sign
# convention of the real spherical harmonics: J' = S J S with S = diag(+-1) and s_m = s_-m (the reference's Z(angle),
# wigner_d.py:16-25, is fixed: cos(m a) on the diagonal, sin(m a) on the anti-diagonal, so a sign that treats +m and -m
# differently would not commute with it), under which every D^l = Z J Z J Z becomes S D^l S.
_SQ3H = 0.8660254037844386
J1 = ((0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, -1.0))
J2 = ((0.0, 0.0, 0.0, -1.0, 0.0), (0.0, 1.0, 0.0, 0.0, 0.0), (0.0, 0.0, -0.5, 0.0, -_SQ3H), (-1.0, 0.0, 0.0, 0.0, 0.0),
      (0.0, 0.0, -_SQ3H, 0.0, 0.5))


def check_j_convention(path_or_list, atol: float = 1e-5) -> dict:
    """Compare a ``J_dense.pt`` (list indexed by degree, as wigner_d.py:8-9,30 uses it) with this build's J_1, J_2.

    Returns ``{1: S1, 2: S2}``: the sign vectors (s_i = s_(n-1-i), centre +1) with ``J_blob = diag(S) J_build diag(S)`` (all +1 = identical
    convention: reference-trained so3 weights can be served as they are).  A non-trivial S means every D^l of this
    build differs from the blob's by ``S D S`` -- the caller must conjugate (or retrain); no S at all raises
    ``ValueError`` (the file is not a J matrix set in the basis wigner_d.py:16-25 fixes)."""
    import itertools
    blob = torch.load(path_or_list, weights_only=True) if isinstance(path_or_list, (str, os.PathLike)) else path_or_list
    out = {}
    for l, mine in ((1, J1), (2, J2)):
        if len(blob) <= l:
            raise ValueError(f"J set has no degree {l}")
        Jb = torch.as_tensor(blob[l], dtype=torch.float64)
        Jm = torch.tensor(mine, dtype=torch.float64)
        n = 2 * l + 1
        if tuple(Jb.shape) != (n, n):
            raise ValueError(f"J_{l} has shape {tuple(Jb.shape)}, expected {(n, n)}")
        found = None
        for half in itertools.product((1.0, -1.0), repeat=l):               # s_i = s_(n-1-i); the overall sign cancels: centre = +1
            S = torch.tensor(half + (1.0,) + half[::-1], dtype=torch.float64)
            if (S[:, None] * Jm * S[None, :] - Jb).abs().max() <= atol:
                found = S
                break
        if found is None:
            raise ValueError(f"J_{l} of the file is not S J S of this build's J_{l} for any sign matrix S "
                             f"(max |J_file - J_build| = {(Jb - Jm).abs().max():.3g})")
        out[l] = found
    return out
