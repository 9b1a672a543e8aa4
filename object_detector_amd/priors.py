"""Prior boxes (`od.pb` in the reference: check_assign.py:21,27).  Host logic: the table itself; device work
(decode_locs, encode_truth) goes through libodhip.so.

reference docs/MODEL.md:23-31: three feature maps (40x40, 20x20, 10x10 at 320 input), 8 prior boxes per cell whose
sizes / aspect ratios come from KMeans over the training boxes measured in units of the grid cell (Euclidean distance).
[BUILD-DEFINED]: frozen default (w,h) table below (no VOC data offline), corner form, normalised [0,1] coordinates,
row order (level, y, x, prior); `fit` re-derives the table from data the way MODEL.md describes.
"""
from __future__ import annotations

import numpy as np

STRIDES = (8, 16, 32)
NUM_PRIORS = 8

DEFAULT_PRIOR_WH = np.array([
    [(1.2, 1.2), (2.0, 2.0), (1.2, 2.4), (2.4, 1.2), (3.2, 3.2), (2.0, 4.0), (4.0, 2.0), (5.0, 5.0)],
    [(1.6, 1.6), (2.5, 2.5), (1.6, 3.2), (3.2, 1.6), (4.0, 4.0), (2.6, 5.2), (5.2, 2.6), (6.0, 6.0)],
    [(2.0, 2.0), (3.0, 3.0), (2.0, 4.0), (4.0, 2.0), (4.5, 4.5), (3.2, 6.4), (6.4, 3.2), (8.0, 8.0)],
], dtype=np.float64)


def level_shapes(input_size):
    H, W = int(input_size[0]), int(input_size[1])
    return [(H // s, W // s) for s in STRIDES]


def make_priors(input_size=(320, 320), prior_wh=DEFAULT_PRIOR_WH) -> np.ndarray:
    """f32 [P,4] (x1,y1,x2,y2) in normalised image coordinates."""
    rows = []
    for (gh, gw), wh in zip(level_shapes(input_size), np.asarray(prior_wh, np.float64)):
        cy = (np.arange(gh, dtype=np.float64) + 0.5) / gh
        cx = (np.arange(gw, dtype=np.float64) + 0.5) / gw
        cyg, cxg = np.meshgrid(cy, cx, indexing="ij")
        half_w = wh[:, 0] / gw / 2
        half_h = wh[:, 1] / gh / 2
        lvl = np.empty((gh, gw, NUM_PRIORS, 4), np.float64)
        lvl[..., 0] = cxg[..., None] - half_w
        lvl[..., 1] = cyg[..., None] - half_h
        lvl[..., 2] = cxg[..., None] + half_w
        lvl[..., 3] = cyg[..., None] + half_h
        rows.append(lvl.reshape(-1, 4))
    return np.concatenate(rows).astype(np.float32)


def fit(bboxes_norm: np.ndarray, input_size=(320, 320), seed=0) -> np.ndarray:
    """KMeans (k=8 per level) over box (w,h) in grid-cell units, Euclidean distance (docs/MODEL.md:29-31).
    Boxes are split over the three levels by size terciles [BUILD-DEFINED].  -> [3,8,2] table."""
    from sklearn.cluster import KMeans
    b = np.asarray(bboxes_norm, np.float64)
    wh = np.stack([b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
    order = np.argsort(wh.prod(1))
    table = []
    for lvl, (gh, gw) in enumerate(level_shapes(input_size)):
        part = wh[order[len(order) * lvl // 3: len(order) * (lvl + 1) // 3]] * np.array([gw, gh])
        km = KMeans(n_clusters=NUM_PRIORS, random_state=seed, n_init=4).fit(part)
        c = km.cluster_centers_
        table.append(c[np.argsort(c.prod(1))])
    return np.asarray(table)
