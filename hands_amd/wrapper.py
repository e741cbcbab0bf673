"""``HandsWrapper`` -- the inference-side shell of the reference's wrapper around ``HandsLight``.

The reference's ``HandsWrapper(GenericWrapper(AbstractPL(LightningModule)))``
(src/models/hands_light/wrapper.py:11-25, src/models/generic/wrapper.py:27-75) is a training
harness; only ``inference`` / ``inference_pose`` sit on the forward path and that is what this
class reproduces: run the model, merge ``inputs.*`` + ``pred.*`` + ``meta_info.*`` into one strict
dict and move every tensor to the CPU (generic/wrapper.py:68-75).  Like the reference it does not
switch the module to eval mode; the HIP path has no train-mode behaviour anyway.
"""
from __future__ import annotations

import torch.nn as nn

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

    def forward(self, inputs, targets=None, meta_info=None, mode="extract"):
        if mode not in ("extract", "vis"):
            raise NotImplementedError("hands_amd.HandsWrapper: only the inference modes are built "
                                      "(training/loss/metrics modes of generic/wrapper.py:77-164 are out of scope)")
        meta_info = dict(meta_info)
        meta_info["mano.faces.r"] = self.model.mano_r.faces      # generic/wrapper.py:93-94 pass-through
        meta_info["mano.faces.l"] = self.model.mano_l.faces
        return self.inference_pose(inputs, meta_info)
