"""Oracle: prior-box assignment + target encoding = od.pb.encode_truth (reference check_assign.py:21,25-27).

Row layout (check_assign.py:25-27): col 0 background, col 1 "assigned" flag, cols 2..2+NC one-hot class, last 4 the
corner-form regression target.  The matching rule is [BUILD-DEFINED] (the reference does not pin it):
  1. prior p takes g* = argmax_g IoU(g,p) (ties: lowest g) when IoU >= pos_thr (0.5); neg_thr (0.4) <= IoU < pos_thr
     -> ignored (all-zero row); else background
  2. each GT g in ascending order force-takes p* = argmax_p IoU(g,p) (ties: lowest p) if that IoU > 0; later g wins
  3. target = ((gt - prior) / [pw,ph,pw,ph]) / loc_scale  (inverse of decode_locs), f32 IEEE ops, no FMA
"""
from __future__ import annotations

import numpy as np

POS_THR = np.float32(0.5)
NEG_THR = np.float32(0.4)


def iou_matrix(gt, priors):
    """f32 [G,P], op-for-op the f32 sequence of od_assign_match."""
    f = np.float32
    a = gt[:, None, :].astype(f)
    c = priors[None, :, :].astype(f)
    ix1 = np.maximum(a[..., 0], c[..., 0]); iy1 = np.maximum(a[..., 1], c[..., 1])
    ix2 = np.minimum(a[..., 2], c[..., 2]); iy2 = np.minimum(a[..., 3], c[..., 3])
    iw = np.maximum(ix2 - ix1, f(0)); ih = np.maximum(iy2 - iy1, f(0))
    inter = iw * ih
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_c = (c[..., 2] - c[..., 0]) * (c[..., 3] - c[..., 1])
    uni = (area_a + area_c) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(uni > 0, inter / uni, f(0))
    return out.astype(f)


def encode_truth(gt_boxes, gt_classes, priors, num_classes=20, pos_thr=POS_THR, neg_thr=NEG_THR, loc_scale=0.1):
    """-> (y f32 [P, 2+NC+4], assigned i32 [P]: GT index / -1 background / -2 ignore)"""
    f = np.float32
    priors = np.asarray(priors, f)
    P = len(priors)
    C = 2 + num_classes + 4
    y = np.zeros((P, C), f)
    assigned = np.full(P, -1, np.int32)
    gt_boxes = np.asarray(gt_boxes, f).reshape(-1, 4)
    G = len(gt_boxes)
    if G:
        iou = iou_matrix(gt_boxes, priors)
        best_g = iou.argmax(0)  # first max = lowest g
        best_iou = iou.max(0)
        has = best_iou > 0
        pos = has & (best_iou >= f(pos_thr))
        ign = has & ~pos & (best_iou >= f(neg_thr))
        assigned[pos] = best_g[pos]
        assigned[ign] = -2
        for g in range(G):
            p = int(iou[g].argmax())  # lowest p on ties
            if iou[g, p] > 0:
                assigned[p] = g
    bg = assigned == -1
    y[bg, 0] = 1
    ps = np.nonzero(assigned >= 0)[0]
    if len(ps):
        g = assigned[ps]
        y[ps, 1] = 1
        cls = np.asarray(gt_classes, np.int64)[g]
        ok = (cls >= 0) & (cls < num_classes)
        y[ps[ok], 2 + cls[ok]] = 1
        pr = priors[ps]
        pw = pr[:, 2] - pr[:, 0]; ph = pr[:, 3] - pr[:, 1]
        size = np.stack([pw, ph, pw, ph], 1)
        y[ps, -4:] = ((gt_boxes[g] - pr) / size) / f(loc_scale)
    return y, assigned
