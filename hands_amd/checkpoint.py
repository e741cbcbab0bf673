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


def load_reference_checkpoint(module: torch.nn.Module, path_or_ckpt, prefix: str = "model.", strict: bool = False):
    """Mirror of scripts_method/train.py:34-37 for ``hands_amd.HandsLight`` / ``HAMER`` / ``HandOccNet``.
    Returns torch's (missing_keys, unexpected_keys) report; MANO buffers stay those of the module's
    asset unless the checkpoint carries same-shaped ones."""
    ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=False) if isinstance(path_or_ckpt, str) \
        else path_or_ckpt
    sd = extract_model_state_dict(ckpt, prefix)
    own = module.state_dict()
    sd = {k: v for k, v in sd.items() if k not in own or tuple(own[k].shape) == tuple(v.shape)}
    return module.load_state_dict(sd, strict=strict)
