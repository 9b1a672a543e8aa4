"""tk.ndimage.save (reference voc_evaluate.py:37)."""
import pathlib

import numpy as np


def save(path, img):
    from PIL import Image
    path = pathlib.Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(np.clip(np.asarray(img), 0, 255).astype(np.uint8)).save(path)


def load(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), np.uint8)
