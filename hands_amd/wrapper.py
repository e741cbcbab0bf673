"""``HandsWrapper`` -- the inference-side shell of the reference's wrapper around ``HandsLight``.

The reference's ``HandsWrapper(GenericWrapper(AbstractPL(LightningModule)))``
(src/models/hands_light/wrapper.py:11-25, src/models/generic/wrapper.py:27-75) is a training
harness; only ``inference`` / ``inference_pose`` sit on the forward path and that is what this
class reproduces: run the model, merge ``inputs.*`` + ``pred.*`` + ``meta_info.*`` into one strict
dict and move every tensor to the CPU (generic/wrapper.py:68-75).  Like the reference it does not
switch the module to eval mode; the HIP path has no train-mode behaviour anyway.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import ManoOut, ManoSide, check, ptr
from .engine import DEFAULT_ENGINE
from .hands_light import DEFAULT_ARGS, HandsLight
from .xdict import xdict


class HandsWrapper(nn.Module):
    def __init__(self, args=None, push_images_fn=None, model: HandsLight | None = None):
        super().__init__()
        args = args if args is not None else DEFAULT_ARGS
        self.args = args
        get = args.get if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
        self.model = model if model is not None else HandsLight(
            backbone=get("backbone", "resnet50"), focal_length=get("focal_length", 1000.0),
            img_res=get("img_res", 224), args=args)
        # the reference wrapper owns its own MANO layers for GT processing (generic/wrapper.py:36-39)
        self.mano_r = self.model.mano_r.mano
        self.mano_l = self.model.mano_l.mano

    def inference_pose(self, inputs, meta_info):
        pred = self.model(inputs, meta_info)
        mydict = xdict()
        mydict.merge(xdict(inputs).prefix("inputs."))
        mydict.merge(pred.prefix("pred."))
        mydict.merge(xdict(meta_info).prefix("meta_info."))
        return mydict.detach()

    def inference(self, inputs, meta_info):
        return self.inference_pose(inputs, meta_info)

    # ---- GT preprocessing on device (src/callbacks/process/process_arctic.py:4-75) ----------------
    @torch.no_grad()
    def process_data(self, targets, meta_info):
        """Adds mano.{joints3d,vertices,cam_t,cam_t.wp,v3d.cam,j3d.cam}.{r,l} to ``targets`` from the GT
        MANO parameters (axis-angle pose (B,48), betas (B,10)) and the annotated camera-space joints
        ``mano.j3d.full.{r,l}``; MANO runs on the same LBS kernels as the prediction path."""
        L = _lib.lib()
        K = meta_info["intrinsics"]
        dev = K.device
        if dev.type != "cuda":
            raise RuntimeError("hands_amd.HandsWrapper runs on a HIP device only (no CPU fallback)")
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        K = f32(K)
        P = self.model.packed(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        img_res = float(getattr(self.model, "img_res", 224))
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        for h, mp in (("r", P["mano_r"]), ("l", P["mano_l"])):
            pose, beta, full = f32(targets[f"mano.pose.{h}"]), f32(targets[f"mano.beta.{h}"]), f32(targets[f"mano.j3d.full.{h}"])
            B = pose.shape[0]
            assert pose.shape == (B, 48) and beta.shape == (B, 10) and full.shape == (B, 21, 3)
            o = {k: new(B, n, 3) for k, n in (("vertices", 778), ("joints3d", 21), ("v3d", 778), ("j3d", 21))}
            j2d, cam_t_unused, one_cam = new(B, 21, 2), new(B, 3), torch.ones(B, 3, device=dev)
            mo = ManoOut(ptr(o["vertices"]), ptr(o["joints3d"]), ptr(o["v3d"]), ptr(o["j3d"]), ptr(j2d), ptr(cam_t_unused))
            side = (ManoSide * 1)(ManoSide(mp["consts"], ptr(mp["blend"].w), ptr(mp["blend"].bias), ptr(pose), ptr(beta),
                                           ptr(one_cam), mo))
            check(L.hands_mano_heads_f32(side, 1, ptr(K), 10, img_res, 0.1, B, 1, stream), "mano_heads (axis-angle)")
            v3d_cam, cam_t, cam_wp = new(B, 778, 3), new(B, 3), new(B, 3)
            check(L.hands_gt_targets_f32(ptr(o["joints3d"]), ptr(o["vertices"]), ptr(full), ptr(K), img_res, ptr(v3d_cam),
                                         ptr(cam_t), ptr(cam_wp), B, 778, stream), "gt_targets")
            for key, val in ((f"mano.joints3d.{h}", o["joints3d"]), (f"mano.vertices.{h}", o["vertices"]),
                             (f"mano.cam_t.{h}", cam_t), (f"mano.cam_t.wp.{h}", cam_wp), (f"mano.v3d.cam.{h}", v3d_cam),
                             (f"mano.j3d.cam.{h}", full)):
                targets.overwrite(key, val)
        return targets

    def _unnormalize(self, t):
        L = _lib.lib()
        t = t.to(dtype=torch.float32).contiguous()
        out = torch.empty_like(t)
        check(L.hands_unnormalize_kp2d_f32(ptr(t), ptr(out), t.numel(), float(getattr(self.model, "img_res", 224)),
                                           torch.cuda.current_stream(t.device).cuda_stream), "unnormalize_kp2d")
        return out

    def forward(self, inputs, targets=None, meta_info=None, mode="test"):
        """GenericWrapper.forward without the training parts (src/models/generic/wrapper.py:77-164):
        GT preprocessing -> model -> 2-D de-normalisation -> metrics (``test``) / merged dict (``vis``,
        ``extract``).  Losses are training-only and out of scope: the loss dict is returned empty."""
        if mode not in ("test", "extract", "vis"):
            raise NotImplementedError("hands_amd.HandsWrapper: the training mode (losses, optimiser) is out of scope")
        inputs, targets, meta_info = xdict(inputs), xdict(targets or {}), xdict(meta_info)
        targets = self.process_data(targets, meta_info)
        meta_info.overwrite("mano.faces.r", self.model.mano_r.faces)      # generic/wrapper.py:93-94
        meta_info.overwrite("mano.faces.l", self.model.mano_l.faces)
        pred = self.model(inputs, meta_info)
        for key in list(pred.keys()):                                      # generic/wrapper.py:118-134
            if "2d.norm" in key:
                assert key in targets.keys(), f"Do not have key {key}"
                dk = key.replace(".norm", "")
                dev = pred[key].device
                pred[dk] = self._unnormalize(pred[key])
                targets[dk] = self._unnormalize(targets[key].to(dev))
        merged = lambda: xdict({**inputs.prefix("inputs."), **pred.prefix("pred."), **targets.prefix("targets."),
                                **meta_info.prefix("meta_info.")}).detach()
        if mode == "vis":
            return merged()
        from .metrics import evaluate_metrics
        metrics_all = evaluate_metrics(pred, targets, meta_info).detach()
        out_dict = xdict()
        out_dict["imgname"] = meta_info.get("imgname")
        out_dict.merge({"metric." + k: v for k, v in metrics_all.items()})
        if mode == "extract":
            return merged()
        return out_dict, {}


class HaMeRWrapper(HandsWrapper):
    """src/models/hamer_light/wrapper.py:5-19: the same shell around ``HAMER``."""

    def __init__(self, args=None, push_images_fn=None, model=None):
        from .hamer import HAMER, HAMER_DEFAULT_ARGS
        args = args if args is not None else HAMER_DEFAULT_ARGS
        get = args.get if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
        if model is None:
            model = HAMER(focal_length=get("focal_length", 1000.0), img_res=get("img_res", 224), args=args)
        super().__init__(args, push_images_fn, model=model)


class HandOccNetWrapper(HandsWrapper):
    """src/models/handoccnet_light/wrapper.py:5-19: the same shell around ``HandOccNet``."""

    def __init__(self, args=None, push_images_fn=None, model=None):
        from .handoccnet import HANDOCC_DEFAULT_ARGS, HandOccNet
        args = args if args is not None else HANDOCC_DEFAULT_ARGS
        get = args.get if hasattr(args, "get") else (lambda k, d=None: getattr(args, k, d))
        if model is None:
            model = HandOccNet(focal_length=get("focal_length", 1000.0), img_res=get("img_res", 224), args=args)
        super().__init__(args, push_images_fn, model=model)
