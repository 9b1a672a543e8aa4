"""Oracle: prior boxes, decode_locs, confidence (numpy f32, op-for-op the sequence the HIP kernels execute).

reference docs/MODEL.md:23-31 (3 maps x 8 priors, sizes relative to the grid cell), :54-58 (conf = objectness x class
probability, plain product), check_assign.py:27 (decode_locs(zeros) == the prior boxes themselves; corner form).
[BUILD-DEFINED]: prior (w,h) table, loc_scale 0.1, priors in normalised [0,1] image coordinates, order (level, y, x, a).
"""
from __future__ import annotations

import numpy as np

LOC_SCALE = np.float32(0.1)
STRIDES = (8, 16, 32)

# (w, h) of the 8 priors of each level, in units of that level's grid cell  [BUILD-DEFINED frozen table]
DEFAULT_PRIOR_WH = np.array([
    [(1.2, 1.2), (2.0, 2.0), (1.2, 2.4), (2.4, 1.2), (3.2, 3.2), (2.0, 4.0), (4.0, 2.0), (5.0, 5.0)],
    [(1.6, 1.6), (2.5, 2.5), (1.6, 3.2), (3.2, 1.6), (4.0, 4.0), (2.6, 5.2), (5.2, 2.6), (6.0, 6.0)],
    [(2.0, 2.0), (3.0, 3.0), (2.0, 4.0), (4.0, 2.0), (4.5, 4.5), (3.2, 6.4), (6.4, 3.2), (8.0, 8.0)],
], dtype=np.float64)


def make_priors(input_size=(320, 320), prior_wh=DEFAULT_PRIOR_WH):
    """-> f32 [P,4] corner-form priors (x1,y1,x2,y2), P = 8*sum(H_l*W_l); computed in f64, rounded once."""
    H, W = input_size
    out = []
    for lvl, s in enumerate(STRIDES):
        gh, gw = H // s, W // s
        ys, xs = np.meshgrid(np.arange(gh, dtype=np.float64), np.arange(gw, dtype=np.float64), indexing="ij")
        cx = ((xs + 0.5) / gw)[..., None]
        cy = ((ys + 0.5) / gh)[..., None]
        pw = (prior_wh[lvl][:, 0] / gw)[None, None, :]
        ph = (prior_wh[lvl][:, 1] / gh)[None, None, :]
        b = np.stack([cx - pw / 2, cy - ph / 2, cx + pw / 2, cy + ph / 2], axis=-1)  # [gh,gw,8,4]
        out.append(b.reshape(-1, 4))
    return np.concatenate(out, 0).astype(np.float32)


def decode_locs(locs, priors, loc_scale=LOC_SCALE, clip=False):
    """boxes = prior + (loc*loc_scale) * [pw,ph,pw,ph]; every op a separately-rounded f32 op (no FMA)."""
    locs = np.asarray(locs, np.float32)
    priors = np.asarray(priors, np.float32)
    pw = priors[:, 2] - priors[:, 0]
    ph = priors[:, 3] - priors[:, 1]
    size = np.stack([pw, ph, pw, ph], axis=-1)
    out = priors + (locs * np.float32(loc_scale)) * size
    if clip:
        out = np.minimum(np.maximum(out, np.float32(0)), np.float32(1))
    return out.astype(np.float32)


def confidence(pred, num_classes=20):
    """pred f32 [...,2+NC+4] -> conf f32 [...,NC] = sigmoid(l1-l0) * softmax(classes), same op order as od_head_post."""
    pred = np.asarray(pred, np.float32)
    one = np.float32(1)
    obj = one / (one + np.exp(pred[..., 0] - pred[..., 1], dtype=np.float32))
    cl = pred[..., 2:2 + num_classes]
    mx = cl.max(axis=-1, keepdims=True)
    e = np.exp(cl - mx, dtype=np.float32)
    s = np.zeros(e.shape[:-1], np.float32)
    for c in range(num_classes):  # sequential f32 sum, c ascending (as the kernel)
        s = s + e[..., c]
    return (obj[..., None] * (e / s[..., None])).astype(np.float32)


def head_postprocess(pred, priors, num_classes=20, loc_scale=LOC_SCALE, clip=True):
    conf = confidence(pred, num_classes)
    B = pred.shape[0]
    boxes = np.stack([decode_locs(pred[b, :, -4:], priors, loc_scale, clip) for b in range(B)])
    return conf, boxes
