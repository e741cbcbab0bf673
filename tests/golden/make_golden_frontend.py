#!/usr/bin/env python3
"""Generate tests/golden/frontend.npz with the REAL reference crop geometry
(`common/data_utils.py:56-91 gen_trans_from_patch_cv`, `:495-509 crop_and_pad`).  Dev container only.

cv2 is absent, so `cv2.warpAffine` is replaced by a recorder that returns zeros (the fixture holds
no pixel data) and `cv2.getAffineTransform` by a 3-point solve in double -- the one stub here that
carries arithmetic (flagged in the metadata).  What the fixture pins: the crop window
`crop_and_pad` returns, the box it hands to the patch generator and the float32 affine that
`gen_trans_from_patch_cv` builds from it.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_shims import *  # noqa: F401,F403
from _ref_shims import META
import numpy as np

from oracle import frontend_oracle as F

cv2 = sys.modules["cv2"]
cv2.INTER_CUBIC = 2
cv2.getAffineTransform = lambda s, d: F.get_affine_transform(s, d)
_calls = []


def _warp(img, trans, dsize, flags=None):
    _calls.append(np.array(trans))
    return np.zeros((dsize[1], dsize[0], img.shape[2]), np.float32)


cv2.warpAffine = _warp
import common.data_utils as du  # noqa: E402  (real reference code)


class Args(dict):
    __getattr__ = dict.get


def main():
    rng = np.random.default_rng(5)
    args = Args(img_res=224, img_res_ds=224)
    img = np.zeros((3, 224, 224), np.float32)
    boxes, scales, new_boxes, transes = [], [], [], []
    cases = [None]
    for _ in range(60):
        x0, y0 = rng.integers(0, 200, 2)
        w, h = rng.integers(1, 224 - max(x0, y0), 2)
        cases.append(np.array([x0, y0, w, h]).astype(np.int16))
    cases += [np.array([0, 0, 223, 223], np.int16), np.array([100, 100, 1, 1], np.int16), np.array([200, 3, 23, 220], np.int16)]
    for i, box in enumerate(cases):
        for scale in (1.5, 2.5):
            _calls.clear()
            _, nb = du.crop_and_pad(img, box, args, scale=scale)
            boxes.append(np.full(4, -1, np.int64) if box is None else box.astype(np.int64))
            scales.append(scale)
            new_boxes.append(np.asarray(nb).astype(np.int64))
            transes.append(_calls[0].astype(np.float32))
    # gen_trans_from_patch_cv on its own, incl. non-square destination and fractional centres
    gt_in, gt_out = [], []
    for _ in range(32):
        c = rng.uniform(0, 224, 2)
        s = rng.uniform(4, 400, 2)
        d = rng.integers(32, 300, 2)
        gt_in.append([c[0], c[1], s[0], s[1], d[0], d[1]])
        gt_out.append(du.gen_trans_from_patch_cv(c[0], c[1], s[0], s[1], d[0], d[1], 1.0, 0.0))
    meta = dict(META, what="common/data_utils.py crop_and_pad (returned window + affine handed to cv2.warpAffine) and "
                "gen_trans_from_patch_cv; cv2.getAffineTransform is an ARITHMETIC STUB (double 3-point solve), cv2.warpAffine a recorder")
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend.npz"),
                        box_xywh=np.stack(boxes), scale=np.array(scales), new_bbox=np.stack(new_boxes), trans=np.stack(transes),
                        gen_trans_in=np.array(gt_in), gen_trans_out=np.stack(gt_out).astype(np.float32),
                        meta=np.array(json.dumps(meta)))
    print("wrote frontend.npz", len(boxes), "crop cases")


if __name__ == "__main__":
    main()
