"""Oracle: candidate selection + class-aware greedy NMS (numpy; integer/index output is the bit-exact target).

reference docs/MODEL.md:78-82: among the confident predictions, overlapping ones of the SAME class keep only the most
confident.  [BUILD-DEFINED] and frozen here (SURVEY.md §7 "bit-exact NMS indices"):
  - candidates: conf > conf_threshold, flat index = p*NC + c
  - total order: (conf desc, flat asc)  == descending u64 key (float_bits(conf) << 32) | (0xFFFFFFFF - flat)
  - pre-NMS top-K (K <= 1024) over all classes of one image, max_det kept boxes
  - suppress iff  inter > thr * ((area_a + area_b) - inter), all f32, one rounding per op, areas (x2-x1)*(y2-y1)
  - strict=True: suppression ignores the class
"""
from __future__ import annotations

import numpy as np


def make_keys(conf_flat, conf_threshold):
    """conf f32 [N] -> (keys u64 [n_cand]) of the candidates, unsorted."""
    conf_flat = np.ascontiguousarray(conf_flat, np.float32)
    idx = np.nonzero(conf_flat > np.float32(conf_threshold))[0].astype(np.uint64)
    bits = conf_flat.view(np.uint32)[idx.astype(np.int64)].astype(np.uint64)
    return (bits << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idx)


def topk_keys(conf_flat, K, conf_threshold):
    """The K largest keys, sorted descending (the set od_topk_scores must produce, in od_nms_sort's order)."""
    keys = make_keys(conf_flat, conf_threshold)
    keys = np.sort(keys)[::-1]
    return keys[:K].copy()


def suppress_matrix_row(a, boxes, thr):
    f = np.float32
    ix1 = np.maximum(a[0], boxes[:, 0]); iy1 = np.maximum(a[1], boxes[:, 1])
    ix2 = np.minimum(a[2], boxes[:, 2]); iy2 = np.minimum(a[3], boxes[:, 3])
    iw = np.maximum(ix2 - ix1, f(0)); ih = np.maximum(iy2 - iy1, f(0))
    inter = iw * ih
    area_a = (a[2] - a[0]) * (a[3] - a[1])
    area_b = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    uni = (area_a + area_b) - inter
    return inter > f(thr) * uni


def nms_image(boxes, sorted_keys, num_classes, iou_threshold=0.45, strict=False, max_det=200):
    """boxes f32 [P,4]; sorted_keys u64 descending -> kept flat indices int32 (rank order, <= max_det)."""
    boxes = np.asarray(boxes, np.float32)
    flat = (np.uint64(0xFFFFFFFF) - (sorted_keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    p = flat // num_classes
    c = flat % num_classes
    cb = boxes[p]
    n = len(flat)
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(int(flat[i]))
        if i + 1 < n:
            sup = suppress_matrix_row(cb[i], cb[i + 1:], iou_threshold)
            if not strict:
                sup &= (c[i + 1:] == c[i])
            removed[i + 1:] |= sup
    return np.asarray(keep[:max_det], np.int32)


def detect_image(conf, boxes, K=1024, conf_threshold=0.01, iou_threshold=0.45, strict=False, max_det=200):
    """conf f32 [P,NC], boxes f32 [P,4] -> (kept flat idx int32, classes, confs, boxes)"""
    P, NC = conf.shape
    keys = topk_keys(conf.reshape(-1), K, conf_threshold)
    keep = nms_image(boxes, keys, NC, iou_threshold, strict, max_det)
    return keep, keep % NC, conf.reshape(-1)[keep], boxes[keep // NC]
