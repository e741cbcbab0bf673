"""CPU restatement (numpy) of the reference's crop / KPE front-end -- SURVEY.md §8 (f2).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in hands_amd/).  The product path is the HIP
kernels `hands_frontend_boxes_f32` / `hands_warp_affine_cubic_norm_f32` behind `hands_amd.frontend`.

What it restates (test-time branch of `HandsLightDataset.__getitem__`, no augmentation, no flip):
  * hand boxes from the GT 2-D joints           src/datasets/hands_light_dataset.py:137-152
  * `crop_and_pad` (box -> square crop window)   common/data_utils.py:495-509
  * `gen_trans_from_patch_cv`                    common/data_utils.py:56-91
  * `generate_patch_image_clean` (cv2.warpAffine, INTER_CUBIC, constant-0 border)
                                                 common/data_utils.py:423-460
  * clip to [0,1] + torchvision `Normalize`      hands_light_dataset.py:171-178,528
  * center / corner KPE angles                   hands_light_dataset.py:256-279

Pinning status
  * box arithmetic, `gen_trans_from_patch_cv`, `crop_and_pad`'s returned window: PINNED -- the real
    `common.data_utils` functions are run (tests/golden/make_golden_frontend.py) with an inert cv2
    stub whose only arithmetic is `getAffineTransform` (3-point solve, flagged in the fixture).
  * KPE angles: the statements are inline in `__getitem__` (needs the ARCTIC data to run); restated
    line by line, cross-checked against `compute_*_pos_enc` inputs only by shape/dtype: parity
    unpinned beyond the formula.
  * `cv2.warpAffine` / `cv2.getAffineTransform`: OpenCV is a third-party dependency that is absent
    here (version unpinned, README.md:20 defers to ARCTIC's environment) -> **parity unpinned**.
    The restatement follows OpenCV's published algorithm (imgwarp.cpp): invert the 2x3 matrix in
    double; fixed-point source coordinates with AB_BITS=10, INTER_BITS=5 (1/32-pixel phases,
    round_delta=16); 32x32 table of separable cubic weights (A=-0.75, float32); 4x4 taps,
    BORDER_CONSTANT value 0.  Pinned by properties: identity transform is an exact copy, integer
    translations are exact shifts, weights sum to 1 (constants are reproduced), the A=-0.75 table
    values at phase 1/2.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
ROUND_DELTA = AB_SCALE // INTER_TAB_SIZE // 2


# ------------------------------------------------------------------------------------------------
# boxes (hands_light_dataset.py:137-152) and crop windows (data_utils.py:495-509)
# ------------------------------------------------------------------------------------------------
def bbox_from_joints2d(j2d_norm, img_res):
    """j2d_norm (21, >=2) float32 in [-1,1] -> ([x0,y0,w,h] int16 or None, bbox_og int array)."""
    j2d_norm = np.asarray(j2d_norm, dtype=np.float32)
    pix = ((j2d_norm[..., :2] + 1) / 2) * (img_res - 1)            # float32
    box = np.array([pix[..., 0].min(), pix[..., 1].min(), pix[..., 0].max(), pix[..., 1].max()]).clip(0, img_res - 1)
    box = np.array([box[0], box[1], box[2] - box[0], box[3] - box[1]]).astype(np.int16)
    og = box.copy()
    if box[2] == 0 or box[3] == 0:
        return None, np.array([0, 0, img_res - 1, img_res - 1])
    return box, og


def crop_window(bbox, img_res, scale):
    """`crop_and_pad`'s geometry: returns (patch box [cx, cy, w, h] for the affine, new_bbox [x0,y0,x1,y1] int16)."""
    if bbox is None:
        return [img_res / 2, img_res / 2, img_res, img_res], np.array([0, 0, img_res - 1, img_res - 1])
    x0, y0, x1, y1 = int(bbox[0]), int(bbox[1]), int(bbox[0] + bbox[2]), int(bbox[1] + bbox[3])
    x_mid, y_mid, width, height = (x0 + x1) // 2, (y0 + y1) // 2, x1 - x0, y1 - y0
    size = max(width, height)
    new_bbox = np.array([x_mid - (size * scale) // 2, y_mid - (size * scale) // 2,
                         x_mid + (size * scale) // 2, y_mid + (size * scale) // 2]).clip(0, img_res - 1).astype(np.int16)
    return [x_mid, y_mid, size * scale, size * scale], new_bbox


def get_affine_transform(src, dst):
    """cv2.getAffineTransform: the 2x3 map with dst_i = M [src_i; 1], solved in double (LU)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[i, 0:2], A[i, 2] = src[i], 1.0
        A[i + 3, 3:5], A[i + 3, 5] = src[i], 1.0
        b[i], b[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def gen_trans_from_patch(c_x, c_y, src_width, src_height, dst_width, dst_height, scale=1.0, rot=0.0):
    """data_utils.py:56-91 with rot in degrees (the front-end always passes 0)."""
    src_w, src_h = src_width * scale, src_height * scale
    src_center = np.array([c_x, c_y], dtype=np.float32)
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)

    def rot2(p):
        return np.array([p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs], dtype=np.float32)

    src_down = rot2(np.array([0, src_h * 0.5], dtype=np.float32))
    src_right = rot2(np.array([src_w * 0.5, 0], dtype=np.float32))
    dst_center = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst_down = np.array([0, dst_height * 0.5], dtype=np.float32)
    dst_right = np.array([dst_width * 0.5, 0], dtype=np.float32)
    src = np.stack([src_center, src_center + src_down, src_center + src_right]).astype(np.float32)
    dst = np.stack([dst_center, dst_center + dst_down, dst_center + dst_right]).astype(np.float32)
    return get_affine_transform(src, dst).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# cv2.warpAffine(..., flags=INTER_CUBIC), float32 image, BORDER_CONSTANT(0)
# ------------------------------------------------------------------------------------------------
def cubic_coeffs(x):
    """OpenCV interpolateCubic, float32 arithmetic, A = -0.75."""
    x = np.float32(x)
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def cubic_table():
    """(32, 4) float32: weights of the 4 taps for every 1/32-pixel phase."""
    scale = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    return np.stack([cubic_coeffs(np.float32(i) * scale) for i in range(INTER_TAB_SIZE)])


def invert_affine(trans):
    """warpAffine's in-place inversion of the forward 2x3 matrix (double)."""
    M = np.asarray(trans, dtype=np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def _sat_int(v):
    return np.clip(np.rint(v), -2147483648, 2147483647).astype(np.int64)


def warp_coords(trans, out_h, out_w):
    """Fixed-point source coordinates: integer tap origin (sx, sy) and 1/32 phases (ax, ay) per output pixel."""
    M = invert_affine(trans)
    x = np.arange(out_w, dtype=np.float64)
    y = np.arange(out_h, dtype=np.float64)
    adelta = _sat_int(M[0] * x * AB_SCALE)
    bdelta = _sat_int(M[3] * x * AB_SCALE)
    X0 = _sat_int((M[1] * y + M[2]) * AB_SCALE) + ROUND_DELTA
    Y0 = _sat_int((M[4] * y + M[5]) * AB_SCALE) + ROUND_DELTA
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    return sx, sy, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def warp_affine_cubic(img_hwc, trans, out_h, out_w):
    """img_hwc (H, W, C) float32 -> (out_h, out_w, C) float32."""
    img = np.asarray(img_hwc, dtype=np.float32)
    H, W, C = img.shape
    tab = cubic_table()
    sx, sy, ax, ay = warp_coords(trans, out_h, out_w)
    sx, sy = sx - 1, sy - 1
    wx, wy = tab[ax], tab[ay]                                     # (oh, ow, 4)
    interior = (sx >= 0) & (sx < max(W - 3, 0)) & (sy >= 0) & (sy < max(H - 3, 0))
    out_int = np.zeros((out_h, out_w, C), np.float32)
    out_brd = np.zeros((out_h, out_w, C), np.float32)
    for i in range(4):
        yy = sy + i
        yok = (yy >= 0) & (yy < H)
        yc = np.clip(yy, 0, H - 1)
        row = None
        for j in range(4):
            xx = sx + j
            ok = yok & (xx >= 0) & (xx < W)
            xc = np.clip(xx, 0, W - 1)
            w = (wy[..., i] * wx[..., j]).astype(np.float32)[..., None]
            s = np.where(ok[..., None], img[yc, xc], np.float32(0))
            term = (s * w).astype(np.float32)
            row = term if row is None else (row + term).astype(np.float32)      # interior: row sums first
            out_brd = (out_brd + term).astype(np.float32)                        # border: one running sum
        out_int = row if i == 0 else (out_int + row).astype(np.float32)
    return np.where(interior[..., None], out_int, out_brd).astype(np.float32)


def generate_patch(img_chw, patch_box, out_res):
    """generate_patch_image_clean(...)[0] + np.clip(0,1), CHW in / CHW out (data_utils.py:423-460,495-509)."""
    trans = gen_trans_from_patch(patch_box[0], patch_box[1], patch_box[2], patch_box[3], out_res, out_res)
    patch = warp_affine_cubic(np.transpose(img_chw, (1, 2, 0)), trans, out_res, out_res)
    return np.clip(patch, 0, 1).transpose(2, 0, 1), trans


def normalize_img(x_chw, mean, std):
    """torchvision Normalize on a float32 CHW image."""
    mean = np.asarray(mean, np.float32)[:, None, None]
    std = np.asarray(std, np.float32)[:, None, None]
    return ((x_chw.astype(np.float32) - mean) / std).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# KPE angles (hands_light_dataset.py:256-279)
# ------------------------------------------------------------------------------------------------
def kpe_angles(bbox_xyxy, K):
    """bbox [x0,y0,x1,y1] (int16), K (3,3) float32 -> (center_angle (2,), corner_angle (8,)) float32.
    dtype note: `center` is float64 (int16/2.0) so its atan2 runs in double; `corner - K` is
    int16 - float32 = float32, so the corner atan2 runs in float32 -- as the reference's numpy does
    with the float32 intrinsics of `get_wp_intrix` (common/data_utils.py:376-385)."""
    b = np.asarray(bbox_xyxy)
    K = np.asarray(K)      # float32 intrinsics, or the float64 stand-in of no_intrx_matrix() (then everything below is float64)
    center = (b[:2] + b[2:]) / 2.0
    center_angle = np.array([np.arctan2(center[0] - K[0, 2], K[0, 0]), np.arctan2(center[1] - K[1, 2], K[1, 1])]).astype(np.float32)
    corner = np.array([[b[0], b[1]], [b[0], b[3]], [b[2], b[1]], [b[2], b[3]]])
    corner = np.stack([corner[:, 0] - K[0, 2], corner[:, 1] - K[1, 2]], axis=-1)
    corner_angle = np.arctan2(corner, np.array([[K[0, 0], K[1, 1]]])).flatten().astype(np.float32)
    return center_angle, corner_angle


def no_intrx_matrix(img_res=224):
    """hands_light_dataset.py:247-253 (args.no_intrx): np.eye(3) with f = c = img_res / 2 -- float64."""
    K = np.eye(3)
    K[0, 0] = K[1, 1] = K[0, 2] = K[1, 2] = img_res / 2
    return K


def dense_maps(bbox_xyxy, K, img_res=224, cam_conv=False):
    """hands_light_dataset.py:281-300 ('dense' / 'dense_latent') and :302-333 ('cam_conv'): per-pixel viewing angles of the crop
    window (first index x: meshgrid 'ij') -- int64 grid minus the float32 intrinsic is float64, atan2 runs in double -- in the
    top-left corner of a zero (img_res, img_res) map, plus the window mask.  -> (angle (2 | 6, R, R), mask (R, R)) float32."""
    b = [int(v) for v in np.asarray(bbox_xyxy)]
    K = np.asarray(K)
    xg, yg = np.meshgrid(range(b[0], b[2] + 1), range(b[1], b[3] + 1), indexing="ij")
    pix = np.stack([xg - K[0, 2], yg - K[1, 2]], axis=-1)
    ang = np.arctan2(pix, np.array([[K[0, 0], K[1, 1]]])).transpose(2, 0, 1).astype(np.float32)
    maps = [ang]
    if cam_conv:
        maps.append(pix.transpose(2, 0, 1).astype(np.float32))
        maps.append(np.stack([2 * xg / img_res - 1, 2 * yg / img_res - 1], axis=-1).transpose(2, 0, 1).astype(np.float32))
    maps = np.concatenate(maps, 0)
    out = np.zeros((maps.shape[0], img_res, img_res))
    out[:, :maps.shape[1], :maps.shape[2]] = maps
    mask = np.zeros((img_res, img_res))
    mask[:maps.shape[1], :maps.shape[2]] = 1
    return out.astype(np.float32), mask.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# whole front-end for one sample (test-time branch)
# ------------------------------------------------------------------------------------------------
def frontend_sample(img01_chw, j2d_r, j2d_l, K, img_res=224, out_res=224, bbox_scale=1.5,
                    mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    out = {}
    for h, j2d in (("r", j2d_r), ("l", j2d_l)):
        box, og = bbox_from_joints2d(j2d, img_res)
        patch_box, new_bbox = crop_window(box, img_res, bbox_scale)
        crop, trans = generate_patch(img01_chw, patch_box, out_res)
        out[f"{h}_img"] = normalize_img(crop, mean, std)
        out[f"{h}_bbox"] = new_bbox
        out[f"{h}_bbox_og"] = og
        out[f"{h}_trans"] = trans
        out[f"{h}_center_angle"], out[f"{h}_corner_angle"] = kpe_angles(new_bbox, K)
    full, _ = generate_patch(img01_chw, [img_res / 2, img_res / 2, img_res, img_res], out_res)
    out["img"] = normalize_img(full, mean, std)
    return out
