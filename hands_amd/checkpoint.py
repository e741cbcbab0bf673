"""Load reference checkpoints into the MI355X modules.

The reference trains under PyTorch-Lightning: ``ModelCheckpoint`` writes ``{"state_dict": {...}}`` whose
keys are prefixed ``model.`` (the wrapper attribute, src/models/generic/wrapper.py:27-40) and also
carry the wrapper's own ``mano_r.*`` / ``mano_l.*`` buffers; warm starts use
``load_state_dict(ckpt["state_dict"], strict=False)`` (scripts_method/train.py:34-37).  HaMeR's
released weights use ``backbone.*`` / ``mano_head.*`` (src/models/hamer_light/model.py:34-44).
"""
from __future__ import annotations

import torch


def extract_model_state_dict(ckpt: dict, prefix: str = "model.") -> dict:
    """``ckpt`` = a loaded Lightning checkpoint (or a bare state_dict).  Returns the entries of the
    wrapped model with ``prefix`` stripped; the wrapper-level MANO buffers are dropped."""
    sd = ckpt.get("state_dict", ckpt)
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v
    if not out:          # already un-prefixed (e.g. a model.state_dict() dump)
        out = {k: v for k, v in sd.items() if not k.startswith(("mano_r.", "mano_l.")) or ".mano." in k}
    return out


def load_reference_checkpoint(module: torch.nn.Module, path_or_ckpt, prefix: str = "model.", strict: bool = False,
                              weights_only: bool = True):
    """Mirror of scripts_method/train.py:34-37 for ``hands_amd.HandsLight`` / ``HAMER`` / ``HandOccNet``.
    Returns torch's (missing_keys, unexpected_keys) report.  Like the reference's
    ``load_state_dict(..., strict=False)`` a SIZE mismatch raises (strict=False only forgives missing /
    unexpected names); the one exception is the ``*.mano.*`` asset buffers, which stay those of the
    module's own asset when the checkpoint's differ in shape (e.g. a checkpoint saved with PCA components).
    A checkpoint none of whose keys match the module raises instead of silently leaving the initial weights.
    ``weights_only=True`` refuses pickled code; Lightning checkpoints that carry arbitrary objects need
    ``weights_only=False`` (only for files you trust)."""
    ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=weights_only) if isinstance(path_or_ckpt, str) \
        else path_or_ckpt
    sd = extract_model_state_dict(ckpt, prefix)
    own = module.state_dict()
    keep, bad = {}, []
    for k, v in sd.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            if ".mano." in k:
                continue                      # asset buffer of another shape: keep the module's asset
            bad.append(f"{k}: checkpoint {tuple(v.shape)} vs module {tuple(own[k].shape)}")
            continue
        keep[k] = v
    if bad:
        raise RuntimeError("hands_amd: size mismatch for " + "; ".join(bad[:8]) + (" ..." if len(bad) > 8 else ""))
    if not any(k in own for k in keep):
        raise RuntimeError("hands_amd: no key of the checkpoint matches the module "
                           f"(first keys: {list(sd)[:3]}, prefix {prefix!r})")
    return module.load_state_dict(keep, strict=strict)
