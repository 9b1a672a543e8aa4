"""tk.ml: per-class precision/recall/F-score at a fixed IoU and box plotting (reference voc_evaluate.py:30-31,36;
check_assign.py:29).  Host-side reporting only -- nothing here is on the device path."""
from __future__ import annotations

import os

import numpy as np


def iou_1xn(box, boxes):
    ix1 = np.maximum(box[0], boxes[:, 0]); iy1 = np.maximum(box[1], boxes[:, 1])
    ix2 = np.minimum(box[2], boxes[:, 2]); iy2 = np.minimum(box[3], boxes[:, 3])
    inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
    a = (box[2] - box[0]) * (box[3] - box[1])
    b = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(a + b - inter > 0, inter / (a + b - inter), 0.0)


def compute_scores(gt, pred, iou_threshold=0.5, num_classes=None):
    """-> (precisions, recalls, fscores, supports), one entry per class; greedy matching by confidence."""
    if num_classes is None:
        mx = [int(a.classes.max()) for a in gt if len(a.classes)] + [int(p.classes.max()) for p in pred if len(p.classes)]
        num_classes = (max(mx) + 1) if mx else 0
    tp = np.zeros(num_classes); fp = np.zeros(num_classes); sup = np.zeros(num_classes, np.int64)
    for a, p in zip(gt, pred):
        for c in a.classes:
            sup[c] += 1
        used = np.zeros(len(a.classes), bool)
        for i in np.argsort(-p.confs, kind="stable"):
            c = int(p.classes[i])
            cand = np.nonzero((a.classes == c) & ~used)[0]
            if len(cand):
                ious = iou_1xn(p.bboxes[i], a.bboxes[cand])
                j = int(ious.argmax())
                if ious[j] >= iou_threshold:
                    used[cand[j]] = True
                    tp[c] += 1
                    continue
            fp[c] += 1
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
        rec = np.where(sup > 0, tp / np.maximum(sup, 1), 0.0)
        f = np.where(prec + rec > 0, 2 * prec * rec / (prec + rec), 0.0)
    return prec, rec, f, sup


def print_scores(precisions, recalls, fscores, supports, class_names=None, print_fn=print):
    n = len(precisions)
    names = list(class_names) if class_names is not None else [str(i) for i in range(n)]
    w = max([len(s) for s in names] + [5])
    print_fn(f"{'':>{w}s}  precision   recall  f1-score  support")
    for i in range(n):
        print_fn(f"{names[i]:>{w}s}  {precisions[i]:9.3f} {recalls[i]:8.3f} {fscores[i]:9.3f} {supports[i]:8d}")
    tot = max(1, int(np.sum(supports)))
    avg = lambda v: float(np.sum(np.asarray(v) * supports) / tot)  # noqa: E731
    print_fn(f"{'avg':>{w}s}  {avg(precisions):9.3f} {avg(recalls):8.3f} {avg(fscores):9.3f} {int(np.sum(supports)):8d}")


def plot_objects(base_image, classes, confs, bboxes, class_names=None):
    """Draw boxes (normalised corner form) on an image (path or array) -> uint8 array."""
    from PIL import Image, ImageDraw
    if isinstance(base_image, (str, os.PathLike)):
        img = Image.open(base_image).convert("RGB")
    else:
        img = Image.fromarray(np.clip(np.asarray(base_image), 0, 255).astype(np.uint8)[..., :3])
    d = ImageDraw.Draw(img)
    Wd, H = img.size
    for i in range(len(bboxes)):
        x1, y1, x2, y2 = (float(v) for v in bboxes[i])
        x1, x2, y1, y2 = min(x1, x2), max(x1, x2), min(y1, y2), max(y1, y2)  # an untrained net can decode inverted boxes
        c = int(classes[i]) if classes is not None else 0
        col = tuple(int(v) for v in (np.array([37, 97, 173]) * (c + 1)) % 200 + 55)
        d.rectangle([x1 * Wd, y1 * H, x2 * Wd, y2 * H], outline=col, width=2)
        label = class_names[c] if class_names is not None else str(c)
        if confs is not None:
            label += f" {float(confs[i]):.2f}"
        d.text((x1 * Wd + 2, y1 * H + 2), label, fill=col)
    return np.asarray(img, np.uint8)
