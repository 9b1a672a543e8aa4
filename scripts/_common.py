"""Shared bits of the four entry-point scripts: repo root on sys.path, synthetic VOC-shaped data when there is no VOC."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def synthetic_dataset(n, size=(375, 500), seed=1):
    """VOC-shaped synthetic images + annotations (SURVEY.md §8d recipe): uint8 arrays, 1..10 boxes per image."""
    from object_detector_amd.pb import ObjectsAnnotation
    rng = np.random.default_rng(seed)
    X, y = [], []
    for _ in range(n):
        X.append(rng.integers(0, 256, size + (3,), dtype=np.uint8))
        k = int(np.clip(1 + rng.poisson(1.5), 1, 10))
        c = rng.uniform(0, 1, (k, 2))
        wh = np.exp(rng.uniform(np.log(0.05), np.log(0.9), (k, 2)))
        b = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)
        y.append(ObjectsAnnotation(None, size[1], size[0], rng.integers(0, 20, k), b))
    return np.array(X, dtype=object), np.array(y, dtype=object)


def make_detector(tk, args, batch_size, input_size=(320, 320), **kw):
    OD = tk.dl.od.ObjectDetector
    if args.synthetic:
        return OD.synthetic(batch_size, input_size, **kw)
    return OD.load_voc(batch_size=batch_size, input_size=input_size, weights=args.weights, **kw)


# ---- "shapes": a learnable VOC-shaped detection task that needs no dataset ------------------------------------------
# VOC07+12 cannot be fetched offline, so the train -> export -> voc_validate loop the reference is accepted by
# (README.md:17,22; voc_validate.py:24-31) is closed on generated images instead: 1..3 colour-coded, lightly textured
# rectangles (one fixed colour per class, five VOC class ids) on a smooth random background, VOC-like image sizes.
SHAPE_CLASSES = (1, 6, 7, 11, 14)  # bicycle, car, cat, dog, person
SHAPE_COLOURS = ((220, 40, 40), (40, 200, 60), (50, 80, 230), (230, 210, 40), (200, 60, 210))


def shapes_dataset(n, seed=0, size_range=(240, 400)):
    """-> (X: object array of uint8 [h,w,3] images, y: object array of ObjectsAnnotation)."""
    from scipy.ndimage import uniform_filter
    from object_detector_amd.pb import ObjectsAnnotation
    rng = np.random.default_rng(seed)
    X, y = [], []
    for _ in range(n):
        h, w = (int(v) for v in rng.integers(size_range[0], size_range[1] + 1, 2))
        gh, gw = h // 32 + 2, w // 32 + 2  # greyish blocks with a mild tint: never as saturated as a class colour
        low = (rng.uniform(70, 180, (gh, gw, 1)) + rng.normal(0, 14, (gh, gw, 3))).astype(np.float32)
        bg = np.repeat(np.repeat(low, 32, 0), 32, 1)[:h, :w]
        bg = uniform_filter(bg, size=(9, 9, 1), mode="nearest")  # smooth blocks: no hard edges in the background
        img = bg + 6.0 * rng.standard_normal((h, w, 3), dtype=np.float32)
        boxes, classes = [], []
        for _obj in range(int(rng.integers(1, 4))):
            for _try in range(20):
                bw, bh = rng.uniform(0.15, 0.6, 2)
                x1, y1 = rng.uniform(0.02, 0.98 - bw), rng.uniform(0.02, 0.98 - bh)
                b = np.array([x1, y1, x1 + bw, y1 + bh])
                if all(_iou(b, o) < 0.15 for o in boxes):
                    break
            else:
                continue
            ci = int(rng.integers(0, len(SHAPE_CLASSES)))
            px = (int(round(b[0] * w)), int(round(b[1] * h)), int(round(b[2] * w)), int(round(b[3] * h)))
            col = np.asarray(SHAPE_COLOURS[ci], np.float64) * rng.uniform(0.85, 1.0)
            img[px[1]:px[3], px[0]:px[2]] = col + 10.0 * rng.standard_normal((px[3] - px[1], px[2] - px[0], 3), dtype=np.float32)
            boxes.append(np.array([px[0] / w, px[1] / h, px[2] / w, px[3] / h]))
            classes.append(SHAPE_CLASSES[ci])
        X.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        y.append(ObjectsAnnotation(None, w, h, classes, np.asarray(boxes, np.float32)))
    Xa = np.empty(n, dtype=object)
    Xa[:] = X
    return Xa, np.array(y, dtype=object)


def _iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def write_voc_layout(vocdevkit_dir, X, y, image_set="test", year=2007, fmt="png"):
    """Write images + annotations as a VOCdevkit directory (JPEGImages / Annotations / ImageSets/Main/<set>.txt) that
    tk.data.voc.load_07_test / load_set read back.  fmt="png" keeps the pixels exact (the files still sit in JPEGImages)."""
    from PIL import Image
    from object_detector_amd.tk.data.voc import CLASS_NAMES
    base = pathlib.Path(vocdevkit_dir) / f"VOC{year}"
    for d in ("Annotations", "JPEGImages", "ImageSets/Main"):
        (base / d).mkdir(parents=True, exist_ok=True)
    ids = []
    for i, (img, a) in enumerate(zip(X, y)):
        name = f"{i:06d}"
        ids.append(name)
        h, w = img.shape[:2]
        Image.fromarray(img).save(base / "JPEGImages" / f"{name}.{fmt}", **({"quality": 95} if fmt == "jpg" else {}))
        objs = ""
        for c, b in zip(a.classes, a.bboxes):  # VOC pixels are 1-based inclusive (tk.data.voc.load_annotation undoes this)
            objs += (f"<object><name>{CLASS_NAMES[int(c)]}</name><difficult>0</difficult><bndbox>"
                     f"<xmin>{int(round(b[0] * w)) + 1}</xmin><ymin>{int(round(b[1] * h)) + 1}</ymin>"
                     f"<xmax>{int(round(b[2] * w))}</xmax><ymax>{int(round(b[3] * h))}</ymax></bndbox></object>\n")
        (base / "Annotations" / f"{name}.xml").write_text(
            f"<annotation><filename>{name}.{fmt}</filename><size><width>{w}</width><height>{h}</height><depth>3</depth>"
            f"</size>\n{objs}</annotation>")
    (base / "ImageSets" / "Main" / f"{image_set}.txt").write_text("\n".join(ids))
    return base
