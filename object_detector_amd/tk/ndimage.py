"""tk.ndimage.save (reference voc_evaluate.py:37)."""
import pathlib

import numpy as np


def save(path, img):
    """Written by the main process only: in a multi-rank job every rank holds all (gathered) predictions and runs the same
    plotting loop; N ranks writing the same JPEG concurrently would corrupt it."""
    from PIL import Image
    from .dl import is_main_process
    if not is_main_process():
        return
    path = pathlib.Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(np.clip(np.asarray(img), 0, 255).astype(np.uint8)).save(path)


def load(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), np.uint8)
