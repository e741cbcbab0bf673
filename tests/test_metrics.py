"""Evaluation metrics: oracle vs the reference's own functions (CPU), HIP kernel vs oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

KEYS = ("mpjpe/ra/h", "mpjpe/pa/ra/r", "mpjpe/pa/ra/l", "mpjpe/pa/ra/h", "mrrpe/r/l", "pix_err/r", "pix_err/l", "pix_err/h")


def _load(golden_dir):
    d = np.load(os.path.join(golden_dir, "eval_metrics.npz"))
    pred = {k[len("in/pred."):]: d[k] for k in d.files if k.startswith("in/pred.")}
    targets = {k[len("in/targets."):]: d[k] for k in d.files if k.startswith("in/targets.")}
    ref = {k[4:]: d[k] for k in d.files if k.startswith("out/")}
    return pred, targets, ref


def test_metrics_oracle_matches_reference(golden_dir):
    pred, targets, ref = _load(golden_dir)
    out = MO.evaluate(pred, targets)
    assert set(KEYS) <= set(ref)
    for k in KEYS:
        assert out[k].shape == ref[k].shape, k
        np.testing.assert_array_equal(np.isnan(out[k]), np.isnan(ref[k]), err_msg=k)
        np.testing.assert_allclose(out[k], ref[k], rtol=1e-5, atol=2e-4, equal_nan=True, err_msg=k)


@pytest.mark.gpu
def test_metrics_hip_vs_reference(golden_dir):
    from hands_amd.metrics import evaluate_metrics
    pred, targets, ref = _load(golden_dir)
    dv = lambda d: {k: torch.from_numpy(v).cuda() for k, v in d.items()}
    out = evaluate_metrics(dv(pred), dv(targets))
    for k in KEYS:
        got = out[k].cpu().numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref[k]), err_msg=k)
        np.testing.assert_allclose(got, ref[k], rtol=2e-5, atol=5e-4, equal_nan=True, err_msg=k)   # mm / px
    # degenerate inputs: identical poses, planar joints, a reflected prediction
    g = torch.Generator().manual_seed(0)
    gt = 0.1 * torch.randn(4, 21, 3, generator=g)
    pr = gt.clone()
    pr[1, :, 2] = 0
    gt[1, :, 2] = 0                      # planar
    pr[2] = gt[2] * torch.tensor([1.0, 1.0, -1.0])   # mirror image: best ROTATION, not reflection
    t = {"mano.j3d.cam.r": gt, "mano.j3d.cam.l": gt, "mano.j2d.r": torch.zeros(4, 21, 2), "mano.j2d.l": torch.zeros(4, 21, 2),
         "is_valid": torch.ones(4), "right_valid": torch.ones(4), "left_valid": torch.ones(4),
         "joints_valid_r": torch.ones(4, 21), "joints_valid_l": torch.ones(4, 21)}
    p = {"mano.j3d.cam.r": pr, "mano.j3d.cam.l": pr, "mano.j2d.r": torch.zeros(4, 21, 2), "mano.j2d.l": torch.zeros(4, 21, 2)}
    out = evaluate_metrics(dv({k: v.numpy() for k, v in p.items()}), dv({k: v.numpy() for k, v in t.items()}))
    ref2 = MO.evaluate({k: v.numpy() for k, v in p.items()}, {k: v.numpy() for k, v in t.items()})
    np.testing.assert_allclose(out["mpjpe/pa/ra/r"].cpu().numpy(), ref2["mpjpe/pa/ra/r"], atol=2e-3)
    assert out["mpjpe/pa/ra/r"][0].item() < 1e-3 and out["mpjpe/pa/ra/r"][2].item() > 1.0
