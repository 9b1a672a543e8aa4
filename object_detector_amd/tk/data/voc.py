"""tk.data.voc: PASCAL VOC 2007 test loader and mAP (reference voc_validate.py:24,29-31; docs/MODEL.md:70-76).

evaluate() returns {"mAP": integrated-curve AP averaged over classes, "mAP_VOC": the VOC2007 11-point metric}
(the two labels logged at voc_validate.py:31; class-mean, not image-mean: docs/MODEL.md:74-76).  Host-side only.
"""
from __future__ import annotations

import pathlib
import xml.etree.ElementTree as ET

import numpy as np

from ...pb import ObjectsAnnotation
from ..ml import iou_1xn

CLASS_NAMES = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
               "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
_CLASS_TO_ID = {n: i for i, n in enumerate(CLASS_NAMES)}


def load_annotation(xml_path, image_dir):
    root = ET.parse(xml_path).getroot()
    size = root.find("size")
    w, h = float(size.find("width").text), float(size.find("height").text)
    classes, bboxes, diffs = [], [], []
    for obj in root.findall("object"):
        name = obj.find("name").text.strip()
        if name not in _CLASS_TO_ID:
            continue
        bb = obj.find("bndbox")
        x1, y1, x2, y2 = (float(bb.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax"))
        classes.append(_CLASS_TO_ID[name])
        bboxes.append([(x1 - 1) / w, (y1 - 1) / h, x2 / w, y2 / h])  # VOC pixels are 1-based inclusive
        d = obj.find("difficult")
        diffs.append(bool(int(d.text)) if d is not None else False)
    path = pathlib.Path(image_dir) / root.find("filename").text
    return ObjectsAnnotation(path, w, h, classes, bboxes, diffs)


def load_set(vocdevkit_dir, year, image_set):
    base = pathlib.Path(vocdevkit_dir) / f"VOC{year}"
    ids = (base / "ImageSets" / "Main" / f"{image_set}.txt").read_text().split()
    y = [load_annotation(base / "Annotations" / f"{i}.xml", base / "JPEGImages") for i in ids]
    X = np.array([a.path for a in y], dtype=object)
    return X, np.array(y, dtype=object)


def load_07_test(vocdevkit_dir):
    """-> (X: array of pathlib.Path, y: array of ObjectsAnnotation) for VOC2007 test (voc_validate.py:24)."""
    return load_set(vocdevkit_dir, 2007, "test")


def average_precision(rec, prec, use_07_metric):
    if use_07_metric:  # 11-point: max precision at recall >= t, t = 0, 0.1, ..., 1.0 (docs/MODEL.md:76)
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = prec[rec >= t].max() if np.any(rec >= t) else 0.0
            ap += p / 11.0
        return float(ap)
    mrec = np.concatenate([[0.0], rec, [1.0]])
    mpre = np.concatenate([[0.0], prec, [0.0]])
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def evaluate(y_true, y_pred, iou_threshold=0.5, num_classes=len(CLASS_NAMES)):
    """PASCAL VOC detection AP per class ('difficult' objects neither count nor penalise), mean over classes."""
    aps, aps07 = [], []
    for c in range(num_classes):
        npos = 0
        gts = []
        for a in y_true:
            m = a.classes == c
            diff = a.difficults[m]
            gts.append((a.bboxes[m], diff, np.zeros(int(m.sum()), bool)))
            npos += int((~diff).sum())
        dets = [(float(p.confs[i]), n, p.bboxes[i]) for n, p in enumerate(y_pred)
                for i in np.nonzero(p.classes == c)[0]]
        dets.sort(key=lambda t: -t[0])
        tp = np.zeros(len(dets)); fp = np.zeros(len(dets))
        for k, (_conf, n, box) in enumerate(dets):
            gb, diff, used = gts[n]
            if len(gb):
                ious = iou_1xn(box, gb)
                j = int(ious.argmax())
                if ious[j] >= iou_threshold:
                    if diff[j]:
                        continue
                    if not used[j]:
                        used[j] = True
                        tp[k] = 1
                        continue
            fp[k] = 1
        if npos == 0:
            continue
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        rec = ctp / npos
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        aps.append(average_precision(rec, prec, False))
        aps07.append(average_precision(rec, prec, True))
    return {"mAP": float(np.mean(aps)) if aps else 0.0, "mAP_VOC": float(np.mean(aps07)) if aps07 else 0.0}
