"""Crop / KPE front-end on device: the step in front of the hot path.

``HandsFrontEnd(args)(img, joints2d_r, joints2d_l, intrinsics)`` is the batched, on-device form of the
test-time branch of ``HandsLightDataset.__getitem__`` (src/datasets/hands_light_dataset.py:137-178,
256-279; no augmentation, no flip): hand boxes from the GT 2-D joints, ``crop_and_pad``
(common/data_utils.py:495-509), the cubic crop-resize (``cv2.warpAffine`` inside
``generate_patch_image_clean``, common/data_utils.py:423-460), clip + ImageNet ``Normalize``, and
the center / corner KPE angles.  It returns the ``inputs`` dict ``HandsLight.forward`` consumes
(``img, r_img, l_img, r_bbox, l_bbox, r_bbox_og, l_bbox_og, {r,l}_center_angle, {r,l}_corner_angle``, and with a
per-pixel encoding -- pos_enc 'dense' / 'dense_latent' / 'cam_conv' -- ``{r,l}_dense_angle, {r,l}_dense_mask``,
hands_light_dataset.py:281-333), all on the device.  ``img`` is the (B,3,res,res) RGB image in [0,1] that ``rgb_processing``
(common/data_utils.py:182-204) produces; decoding and that first full-frame crop stay with the caller.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr

IMG_NORM_MEAN = (0.485, 0.456, 0.406)      # src/parsers/parser.py:45-46
IMG_NORM_STD = (0.229, 0.224, 0.225)


class HandsFrontEnd:
    def __init__(self, args=None):
        g = (lambda k, d: d) if args is None else (lambda k, d: d if args.get(k, None) is None else args.get(k))
        self.img_res = int(g("img_res", 224))
        self.img_res_ds = int(g("img_res_ds", 224))
        self.bbox_scale = float(g("bbox_scale", 1.5))                     # hands_light_dataset.py:164-166
        self.mean = tuple(float(v) for v in g("img_norm_mean", IMG_NORM_MEAN))
        self.std = tuple(float(v) for v in g("img_norm_std", IMG_NORM_STD))
        pos_enc = g("pos_enc", "center+corner_latent")
        # per-pixel maps of the crop windows (hands_light_dataset.py:281-333): 2 angle maps, or 6 maps for 'cam_conv'
        self.dense_channels = 0 if pos_enc is None else 6 if "cam_conv" in pos_enc else 2 if "dense" in pos_enc else 0
        self.no_intrx = bool(g("no_intrx", False))      # hands_light_dataset.py:247-253: encodings from f = c = img_res / 2

    def boxes(self, joints2d_r, joints2d_l, intrinsics):
        """-> dict with {r,l}_bbox (B,4) int16 [x0,y0,x1,y1], {r,l}_bbox_og, {r,l}_trans (B,6), angles."""
        L = _lib.lib()
        dev = joints2d_r.device
        if dev.type != "cuda":
            raise RuntimeError("hands_amd.HandsFrontEnd runs on a HIP device only (no CPU fallback)")
        f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        jr, jl, K = f(joints2d_r), f(joints2d_l), f(intrinsics)
        B, ld = jr.shape[0], jr.shape[2]
        assert jr.shape == (B, 21, ld) and jl.shape == (B, 21, ld) and ld >= 2 and K.shape == (B, 3, 3)
        i32 = lambda: torch.empty(B, 4, dtype=torch.int32, device=dev)
        o = {"r_bbox": i32(), "l_bbox": i32(), "r_bbox_og": i32(), "l_bbox_og": i32()}
        for h in "rl":
            o[f"{h}_trans"] = torch.empty(B, 6, device=dev)
            o[f"{h}_center_angle"] = torch.empty(B, 2, device=dev)
            o[f"{h}_corner_angle"] = torch.empty(B, 8, device=dev)
        Kenc = None if self.no_intrx else ptr(K)
        check(L.hands_frontend_boxes_f32(ptr(jr), ptr(jl), ld, Kenc, B, self.img_res, self.img_res_ds, self.bbox_scale,
                                         ptr(o["r_bbox"]), ptr(o["l_bbox"]), ptr(o["r_bbox_og"]), ptr(o["l_bbox_og"]),
                                         ptr(o["r_trans"]), ptr(o["l_trans"]), ptr(o["r_center_angle"]), ptr(o["l_center_angle"]),
                                         ptr(o["r_corner_angle"]), ptr(o["l_corner_angle"]),
                                         torch.cuda.current_stream(dev).cuda_stream), "hands_frontend_boxes_f32")
        if self.dense_channels:
            R, n = self.img_res, self.dense_channels
            for h in "rl":
                o[f"{h}_dense_angle"] = torch.empty(B, n, R, R, device=dev)
                o[f"{h}_dense_mask"] = torch.empty(B, R, R, device=dev)
                check(L.hands_frontend_dense_maps_f32(ptr(o[f"{h}_bbox"]), Kenc, ptr(o[f"{h}_dense_angle"]), ptr(o[f"{h}_dense_mask"]),
                                                      B, R, n, torch.cuda.current_stream(dev).cuda_stream), "hands_frontend_dense_maps_f32")
        for k in ("r_bbox", "l_bbox", "r_bbox_og", "l_bbox_og"):
            o[k] = o[k].to(torch.int16)                                   # the reference's dtype (astype(np.int16))
        return o

    def warp(self, img, trans, out_res=None):
        """cv2.warpAffine(INTER_CUBIC) + clip + Normalize; img (B,3,H,W) in [0,1], trans (B,6) or None (identity)."""
        L = _lib.lib()
        dev = img.device
        if dev.type != "cuda":
            raise RuntimeError("hands_amd.HandsFrontEnd runs on a HIP device only (no CPU fallback)")
        img = img.to(dtype=torch.float32).contiguous()
        B, Cc, H, W = img.shape
        assert Cc == 3
        R = self.img_res_ds if out_res is None else int(out_res)
        out = torch.empty(B, 3, R, R, device=dev)
        mean, std = (C.c_float * 3)(*self.mean), (C.c_float * 3)(*self.std)
        if trans is not None:
            trans = trans.to(device=dev, dtype=torch.float32).contiguous()
            assert trans.shape == (B, 6)
        check(L.hands_warp_affine_cubic_norm_f32(ptr(img), ptr(trans) if trans is not None else None, ptr(out), B, H, W, R, R,
                                                 mean, std, torch.cuda.current_stream(dev).cuda_stream),
              "hands_warp_affine_cubic_norm_f32")
        return out

    def __call__(self, img, joints2d_r, joints2d_l, intrinsics):
        o = self.boxes(joints2d_r, joints2d_l, intrinsics)
        inputs = {"r_img": self.warp(img, o["r_trans"]), "l_img": self.warp(img, o["l_trans"])}
        if self.img_res_ds == self.img_res:
            inputs["img"] = self.warp(img, None)          # identity map: exact copy, then clip + Normalize
        else:
            s = self.img_res_ds / self.img_res
            t = torch.tensor([s, 0, 0, 0, s, 0], dtype=torch.float32, device=img.device).repeat(img.shape[0], 1)
            inputs["img"] = self.warp(img, t)
        for k in ("r_bbox", "l_bbox", "r_bbox_og", "l_bbox_og", "r_center_angle", "l_center_angle",
                  "r_corner_angle", "l_corner_angle") + (("r_dense_angle", "l_dense_angle", "r_dense_mask", "l_dense_mask")
                                                         if self.dense_channels else ()):
            inputs[k] = o[k]
        return inputs
