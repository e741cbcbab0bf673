"""Evaluation metrics on device: the consumer of the gathered predictions.

``evaluate_metrics(pred, targets)`` mirrors the reference's metric functions for the `*_light`
models -- ``eval_mpjpe_ra``, ``eval_mpjpe_pa_ra`` (21-joint branch), ``eval_mrrpe_hand``,
``eval_pixel_error`` (src/utils/eval_modules.py:97-134,221-343,386-428) -- with one kernel launch for
the whole batch instead of a per-sample numpy loop with a LAPACK SVD per hand.  Same keys, same units
(mm / px), same NaN conventions; tensors stay on the device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import EvalIn, EvalOut, check, ptr
from .xdict import xdict


def evaluate_metrics(pred: dict, targets: dict, meta_info: dict | None = None) -> xdict:
    L = _lib.lib()
    dev = pred["mano.j3d.cam.r"].device
    if dev.type != "cuda":
        raise RuntimeError("hands_amd.evaluate_metrics runs on a HIP device only (no CPU fallback)")
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    B = pred["mano.j3d.cam.r"].shape[0]
    keep = [f(pred["mano.j3d.cam.r"]), f(pred["mano.j3d.cam.l"]), f(targets["mano.j3d.cam.r"]), f(targets["mano.j3d.cam.l"]),
            f(pred["mano.j2d.r"]), f(pred["mano.j2d.l"]), f(targets["mano.j2d.r"]), f(targets["mano.j2d.l"]),
            f(targets["is_valid"]), f(targets["right_valid"]), f(targets["left_valid"]),
            f(targets["joints_valid_r"]), f(targets["joints_valid_l"])]
    assert keep[0].shape == (B, 21, 3) and keep[4].shape == (B, 21, 2) and keep[11].shape == (B, 21)
    o = {k: torch.empty(B, device=dev) for k in ("mpjpe/ra/h", "mpjpe/pa/ra/r", "mpjpe/pa/ra/l", "mpjpe/pa/ra/h", "mrrpe/r/l")}
    o["pix_err/r"], o["pix_err/l"] = torch.empty(B, 21, device=dev), torch.empty(B, 21, device=dev)
    ein = EvalIn(*[ptr(t) for t in keep])
    eout = EvalOut(ptr(o["mpjpe/ra/h"]), ptr(o["mpjpe/pa/ra/r"]), ptr(o["mpjpe/pa/ra/l"]), ptr(o["mpjpe/pa/ra/h"]),
                   ptr(o["mrrpe/r/l"]), ptr(o["pix_err/r"]), ptr(o["pix_err/l"]))
    check(L.hands_eval_metrics_f32(C.byref(ein), C.byref(eout), B, torch.cuda.current_stream(dev).cuda_stream),
          "hands_eval_metrics_f32")
    out = xdict(o)
    out["pix_err/h"] = torch.cat((o["pix_err/r"], o["pix_err/l"]), dim=1)
    return out
