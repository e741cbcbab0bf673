"""CPU oracle for the evaluation metrics consuming the predictions.
TEST INFRASTRUCTURE ONLY.  numpy restatement of src/utils/eval_modules.py:97-134 (eval_mpjpe_ra),
:136-219,:320-343 (Procrustes-aligned MPJPE, 21-joint branch), :386-407 (eval_mrrpe_hand),
:410-428 (eval_pixel_error) and common/metrics.py:23-55; pinned by tests/golden/eval_metrics.npz,
which was produced by those reference functions themselves.
"""
import numpy as np


def _nanmean2(a, b):
    """common/torch_utils.py:14-19 over a pair: sum of non-nan / count of non-nan (0/0 -> nan)."""
    st = np.stack([a, b], axis=1).astype(np.float32)
    nan = np.isnan(st)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(nan, 0, st).sum(1) / (~nan).sum(1).astype(np.float32)


def joint3d_error(gt, pred, valid):
    """common/metrics.py:23-32: per-joint L2, rows of invalid hands -> nan."""
    d = np.sqrt(((gt - pred) ** 2).sum(axis=2))
    d[valid.reshape(-1) == 0, :] = np.nan
    return d


def similarity_transform(S1, S2):
    """eval_modules.py:136-187 on (N,3) inputs (the reference transposes to 3xN internally)."""
    S1, S2 = S1.T, S2.T
    mu1, mu2 = S1.mean(axis=1, keepdims=True), S2.mean(axis=1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = np.sum(X1 ** 2)
    K = X1.dot(X2.T)
    U, s, Vh = np.linalg.svd(K)
    V = Vh.T
    Z = np.eye(3)
    Z[-1, -1] *= np.sign(np.linalg.det(U.dot(V.T)))
    R = V.dot(Z.dot(U.T))
    scale = np.trace(R.dot(K)) / var1
    t = mu2 - scale * (R.dot(mu1))
    return (scale * R.dot(S1) + t).T


def evaluate(pred, targets):
    g = lambda d, k: np.asarray(d[k], dtype=np.float32)
    gr, gl, pr, pl = g(targets, "mano.j3d.cam.r"), g(targets, "mano.j3d.cam.l"), g(pred, "mano.j3d.cam.r"), g(pred, "mano.j3d.cam.l")
    is_valid = g(targets, "is_valid")
    lv, rv = g(targets, "left_valid") * is_valid, g(targets, "right_valid") * is_valid
    ra = lambda x: x - x[:, :1, :]
    gr_ra, gl_ra, pr_ra, pl_ra = ra(gr), ra(gl), ra(pr), ra(pl)
    out = {}
    out["mpjpe/ra/h"] = _nanmean2(joint3d_error(gr_ra, pr_ra, rv).mean(axis=1), joint3d_error(gl_ra, pl_ra, lv).mean(axis=1)) * 1000.0

    def pa(gt, pd, valid):
        return np.array([np.mean(np.sqrt(np.sum((gt[i] - similarity_transform(pd[i], gt[i])) ** 2, axis=1))) * valid[i]
                         for i in range(len(gt))])

    par, pal = pa(gr_ra, pr_ra, rv), pa(gl_ra, pl_ra, lv)
    out["mpjpe/pa/ra/r"], out["mpjpe/pa/ra/l"] = par * 1000.0, pal * 1000.0
    out["mpjpe/pa/ra/h"] = _nanmean2(par, pal) * 1000.0
    rel = np.sqrt((((pl[:, 0] - pr[:, 0]) - (gl[:, 0] - gr[:, 0])) ** 2).sum(axis=1))
    rel[(lv * rv) == 0] = np.nan
    out["mrrpe/r/l"] = rel * 1000.0
    for h, v in (("r", rv), ("l", lv)):
        jv = g(targets, "joints_valid_" + h) * v.reshape(-1, 1)
        d = np.sqrt(((g(targets, "mano.j2d." + h) - g(pred, "mano.j2d." + h)) ** 2).sum(axis=2))
        d[jv == 0] = np.nan
        out["pix_err/" + h] = d
    out["pix_err/h"] = np.concatenate((out["pix_err/r"], out["pix_err/l"]), axis=1)
    return out
