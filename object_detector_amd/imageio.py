"""Host-side image decode + resize for `ObjectDetector.predict` (numpy + PIL only: this module is what the optional decode
worker PROCESSES import, so it must not pull in torch or touch the GPU).

reference: the image loading inside `tk.dl.od.ObjectDetector.predict` (voc_validate.py:27: X is a list of file paths);
`keep_aspect=False` is a plain resize to the network input, `True` letterboxes into the top-left corner.
"""
from __future__ import annotations

import os

import numpy as np


def load_image(x, size_hw, keep_aspect=False, return_scale=False):
    """path / ndarray -> uint8 [H,W,3] resized to the network input (host side: PIL).
    return_scale: also the (sx, sy) fraction of the canvas the image occupies (1, 1 unless keep_aspect letterboxes)."""
    out, scale = _load_image(x, size_hw, keep_aspect)
    return (out, scale) if return_scale else out


def _load_image(x, size_hw, keep_aspect):
    from PIL import Image
    H, Wd = size_hw
    if isinstance(x, (str, os.PathLike)):
        img = Image.open(x).convert("RGB")
    else:
        a = np.asarray(x)
        if a.dtype != np.uint8:
            a = np.clip(a, 0, 255).astype(np.uint8)
        if a.shape[:2] == (H, Wd):
            return a[..., :3], (1.0, 1.0)
        img = Image.fromarray(a[..., :3])
    if not keep_aspect:
        return np.asarray(img.resize((Wd, H), Image.BILINEAR), np.uint8), (1.0, 1.0)
    s = min(Wd / img.width, H / img.height)
    nw, nh = max(1, round(img.width * s)), max(1, round(img.height * s))
    canvas = np.zeros((H, Wd, 3), np.uint8)
    canvas[:nh, :nw] = np.asarray(img.resize((nw, nh), Image.BILINEAR), np.uint8)
    return canvas, (nw / Wd, nh / H)


_SHM = {}  # worker process: attached staging blocks by name


def decode_into_shm(shm_name, offset, x, size_hw, keep_aspect):
    """Worker-process entry: decode + resize `x` and write the uint8 [H,W,3] result at `offset` of the shared staging block;
    returns the letterbox scale.  The parent uploads the block to the GPU; nothing here touches the device."""
    from multiprocessing import shared_memory
    shm = _SHM.get(shm_name)
    if shm is None:
        shm = _SHM[shm_name] = shared_memory.SharedMemory(name=shm_name)
    img, sc = _load_image(x, size_hw, keep_aspect)
    H, Wd = size_hw
    dst = np.ndarray((H, Wd, 3), np.uint8, buffer=shm.buf, offset=offset)
    np.copyto(dst, img)
    return sc


def decode_chunk_into_shm(shm_name, offsets, xs, size_hw, keep_aspect):
    """Several images per task: a task costs the parent ~0.15 ms of queueing / pickling, a decode ~1.3 ms."""
    return [decode_into_shm(shm_name, o, x, size_hw, keep_aspect) for o, x in zip(offsets, xs)]
