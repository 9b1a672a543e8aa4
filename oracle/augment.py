"""Oracle: pixel augmentation of the training generator (reference check_generator.py:17-18; docs/MODEL.md:60-64),
numpy f32, op for op the sequence of od_augment_k (csrc/augment.hip) => bit-exact.  TEST INFRASTRUCTURE ONLY.
[BUILD-DEFINED]: crop -> bilinear resize (half-pixel centres, edge clamp) -> flip -> saturation -> contrast ->
brightness -> round half up -> Random-Erasing rectangles on normalised output coordinates."""
from __future__ import annotations

import numpy as np


def augment(img, out_hw, crop=(0.0, 0.0, 1.0, 1.0), flip=False, brightness=0.0, contrast=1.0, saturation=1.0, erase=()):
    f = np.float32
    H, W = out_hw
    sh, sw = img.shape[:2]
    x = np.arange(W, dtype=f)
    y = np.arange(H, dtype=f)
    u = (x + f(0.5)) / f(W)
    v = (y + f(0.5)) / f(H)
    cu, cv = u.copy(), v.copy()
    if flip:
        u = f(1.0) - u
    x1, y1, x2, y2 = (f(c) for c in crop)
    sx = (x1 + u * (x2 - x1)) * f(sw) - f(0.5)
    sy = (y1 + v * (y2 - y1)) * f(sh) - f(0.5)
    sx = np.minimum(np.maximum(sx, f(0)), f(sw - 1))
    sy = np.minimum(np.maximum(sy, f(0)), f(sh - 1))
    x0 = np.floor(sx).astype(np.int64); y0 = np.floor(sy).astype(np.int64)
    x1i = np.minimum(x0 + 1, sw - 1); y1i = np.minimum(y0 + 1, sh - 1)
    fx = (sx - x0.astype(f))[None, :, None]
    fy = (sy - y0.astype(f))[:, None, None]
    im = img[..., :3].astype(f)
    p00 = im[y0][:, x0]; p01 = im[y0][:, x1i]; p10 = im[y1i][:, x0]; p11 = im[y1i][:, x1i]
    top = p00 + (p01 - p00) * fx
    bot = p10 + (p11 - p10) * fx
    c = top + (bot - top) * fy
    gray = (c[..., 0] * f(0.299) + c[..., 1] * f(0.587)) + c[..., 2] * f(0.114)
    t = gray[..., None] + (c - gray[..., None]) * f(saturation)
    t = (t - f(127.5)) * f(contrast) + f(127.5)
    t = t + f(brightness)
    out = np.minimum(np.maximum(np.floor(t + f(0.5)), f(0)), f(255)).astype(np.uint8)
    for (ex1, ey1, ex2, ey2), rgb in erase:
        mx = (cu >= f(ex1)) & (cu < f(ex2))
        my = (cv >= f(ey1)) & (cv < f(ey2))
        out[np.ix_(my, mx)] = np.asarray(rgb, np.uint8)
    return out
