"""Oracle: detection loss + gradient (reference docs/MODEL.md:33-52), numpy f64 so that it also serves as the
high-precision check of the f32 kernel; tests cross-check the gradient by finite differences.

  objectness: 2-class softmax focal loss, alpha_t = alpha (object) / 1-alpha (background), gamma   [:33-37]
  class:      softmax cross-entropy on assigned priors                                              [:39-44]
  box:        smooth-L1 (beta 1; north_star) or MSE = mean_4 d^2 (reference doc) on assigned priors  [:46-52]
  total = (w_obj*sum_obj + w_cls*sum_cls + w_box*sum_box) / max(1, #assigned)
[BUILD-DEFINED]: alpha 0.25, gamma 2 (RetinaNet paper cited at :37), unit weights, normaliser, all-zero row = ignore.
"""
from __future__ import annotations

import numpy as np


def loss_and_grad(pred, y, num_classes=20, alpha=0.25, gamma=2.0, box_mode="smooth_l1", w=(1.0, 1.0, 1.0)):
    pred = np.asarray(pred, np.float64)
    y = np.asarray(y, np.float64)
    NC = num_classes
    grad = np.zeros_like(pred)
    t0, t1 = y[..., 0], y[..., 1]
    pos = t1 > 0.5
    active = (t0 + t1) > 0
    n = max(1, int(pos.sum()))
    # objectness
    l = pred[..., :2]
    m = l.max(-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(l - m).sum(-1))
    lp = l - lse[..., None]
    p = np.exp(lp)
    lpt = np.where(pos, lp[..., 1], lp[..., 0])
    pt = np.exp(lpt)
    a = np.where(pos, alpha, 1 - alpha)
    om = 1 - pt
    l_obj = np.where(active, -a * om ** gamma * lpt, 0.0)
    dl = -a * (om ** gamma - gamma * om ** (gamma - 1) * pt * lpt)
    onehot = np.stack([~pos, pos], -1).astype(np.float64)
    grad[..., :2] = np.where(active[..., None], dl[..., None] * (onehot - p), 0.0) * w[0] / n
    # class
    cl = pred[..., 2:2 + NC]
    mx = cl.max(-1, keepdims=True)
    lse_c = mx[..., 0] + np.log(np.exp(cl - mx).sum(-1))
    lq = cl - lse_c[..., None]
    tc = y[..., 2:2 + NC]
    l_cls = np.where(pos, -(tc * lq).sum(-1), 0.0)
    grad[..., 2:2 + NC] = np.where(pos[..., None], np.exp(lq) - tc, 0.0) * w[1] / n
    # box
    d = pred[..., -4:] - y[..., -4:]
    if box_mode == "smooth_l1":
        ad = np.abs(d)
        lb = np.where(ad < 1, 0.5 * d * d, ad - 0.5)
        gb = np.where(ad < 1, d, np.sign(d))
    else:
        lb = 0.25 * d * d
        gb = 0.5 * d
    l_box = np.where(pos, lb.sum(-1), 0.0)
    grad[..., -4:] = np.where(pos[..., None], gb, 0.0) * w[2] / n
    losses = np.array([l_obj.sum() * w[0] / n, l_cls.sum() * w[1] / n, l_box.sum() * w[2] / n])
    return np.append(losses, losses.sum()), grad
