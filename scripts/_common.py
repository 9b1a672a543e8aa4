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
