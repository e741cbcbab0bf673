"""Manifest-driven parameter containers.

``hands_amd/manifests/<model>.json`` lists the reference model's ``state_dict`` entries (name, shape,
dtype, parameter-or-buffer -- data only, produced by tests/golden/make_golden_*.py).  ``build_tree``
materialises exactly that tree of ``nn.Module`` / ``nn.Parameter`` / buffers under a root module, so a
reference checkpoint loads by name without one hand-written container class per reference module.
The modules are never called: compute lives in libhands_hip.so.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_manifest(name: str) -> dict:
    with open(os.path.join(_HERE, "manifests", name + ".json")) as fh:
        return json.load(fh)


def build_tree(root: nn.Module, manifest: dict, skip_prefixes=()) -> nn.Module:
    for key in sorted(manifest):
        if key.startswith(tuple(skip_prefixes)):
            continue
        info = manifest[key]
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        dtype = getattr(torch, info["dtype"])
        leaf = parts[-1]
        if leaf == "running_var":
            t = torch.ones(info["shape"], dtype=dtype)
        else:
            t = torch.zeros(info["shape"], dtype=dtype)
        if info["kind"] == "param":
            mod.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
        else:
            mod.register_buffer(leaf, t)
    return root
