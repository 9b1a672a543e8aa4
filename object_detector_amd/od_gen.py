"""`tk.dl.od.od_gen`: training data generator (reference check_generator.py:17-18, check_assign.py:21-22).

    gen = create_generator((512, 512), preprocess_input=lambda x: x, encode_truth=od.pb.encode_truth)
    g, steps = gen.flow(X, y, data_augmentation=True)      # infinite iterator of (X_batch, y_batch)

Augmentation list is [BUILD-DEFINED] (docs/MODEL.md:60-64 names Random Erasing "and anything else at hand", erasing
constrained so that boxes are not hidden too much): horizontal flip, random crop/zoom that keeps every box centre,
brightness / contrast / saturation jitter, Random Erasing limited to <= 40 % of any box.  Augmentation PARAMETERS are
sampled on the host (numpy Generator, seeded); the pixel work runs on the device through od_augment_batch when a GPU
is present (`device=`), prior-box encoding always runs on the device (encode_truth = od.pb.encode_truth).
"""
from __future__ import annotations

import math

import numpy as np

from .pb import ObjectsAnnotation


class AugParams:
    """Per-image augmentation parameters (sampled on the host, consumed by the pixel kernel / host path)."""
    __slots__ = ("crop", "flip", "brightness", "contrast", "saturation", "erase")

    def __init__(self):
        self.crop = (0.0, 0.0, 1.0, 1.0)  # x1,y1,x2,y2 in normalised source coordinates
        self.flip = False
        self.brightness = 0.0   # additive, in [0,255] units
        self.contrast = 1.0
        self.saturation = 1.0
        self.erase = []         # [(x1,y1,x2,y2 in normalised OUTPUT coords, (r,g,b))]


def sample_params(rng: np.random.Generator, ann: ObjectsAnnotation, random_erasing=True) -> AugParams:
    p = AugParams()
    b = ann.bboxes
    if rng.random() < 0.5 and len(b):  # crop/zoom keeping all box centres inside
        cx = (b[:, 0] + b[:, 2]) / 2
        cy = (b[:, 1] + b[:, 3]) / 2
        x1 = rng.uniform(0, max(1e-6, min(cx.min(), 0.3)))
        y1 = rng.uniform(0, max(1e-6, min(cy.min(), 0.3)))
        x2 = rng.uniform(min(1 - 1e-6, max(cx.max(), 0.7)), 1)
        y2 = rng.uniform(min(1 - 1e-6, max(cy.max(), 0.7)), 1)
        p.crop = (float(x1), float(y1), float(x2), float(y2))
    p.flip = bool(rng.random() < 0.5)
    p.brightness = float(rng.uniform(-32, 32)) if rng.random() < 0.5 else 0.0
    p.contrast = float(rng.uniform(0.6, 1.4)) if rng.random() < 0.5 else 1.0
    p.saturation = float(rng.uniform(0.6, 1.4)) if rng.random() < 0.5 else 1.0
    if random_erasing and rng.random() < 0.5:
        nb = transform_boxes(b, p)
        for _ in range(int(rng.integers(1, 4))):
            for _try in range(10):
                area = rng.uniform(0.02, 0.2)
                ar = math.exp(rng.uniform(math.log(0.3), math.log(1 / 0.3)))
                w, h = min(1.0, math.sqrt(area * ar)), min(1.0, math.sqrt(area / ar))
                x1, y1 = rng.uniform(0, 1 - w), rng.uniform(0, 1 - h)
                r = np.array([x1, y1, x1 + w, y1 + h], np.float32)
                if _hidden_fraction(r, nb) <= 0.4:  # box-aware constraint
                    p.erase.append((tuple(float(v) for v in r), tuple(int(v) for v in rng.integers(0, 256, 3))))
                    break
    return p


def _hidden_fraction(rect, boxes):
    if not len(boxes):
        return 0.0
    iw = np.clip(np.minimum(rect[2], boxes[:, 2]) - np.maximum(rect[0], boxes[:, 0]), 0, None)
    ih = np.clip(np.minimum(rect[3], boxes[:, 3]) - np.maximum(rect[1], boxes[:, 1]), 0, None)
    area = np.maximum((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), 1e-12)
    return float((iw * ih / area).max())


def transform_boxes(bboxes, p: AugParams):
    """Boxes follow the crop then the flip; clipped to the frame."""
    b = np.asarray(bboxes, np.float32).reshape(-1, 4).copy()
    x1, y1, x2, y2 = p.crop
    b[:, [0, 2]] = (b[:, [0, 2]] - x1) / (x2 - x1)
    b[:, [1, 3]] = (b[:, [1, 3]] - y1) / (y2 - y1)
    if p.flip:
        b[:, [0, 2]] = 1.0 - b[:, [2, 0]]
    return np.clip(b, 0.0, 1.0)


def apply_pixels_host(img_u8: np.ndarray, p: AugParams, out_hw) -> np.ndarray:
    """Host pixel path (PIL resize): used when no device is given."""
    from PIL import Image
    H, W = out_hw
    h0, w0 = img_u8.shape[:2]
    x1, y1, x2, y2 = p.crop
    box = (int(round(x1 * w0)), int(round(y1 * h0)), max(int(round(x2 * w0)), int(round(x1 * w0)) + 1),
           max(int(round(y2 * h0)), int(round(y1 * h0)) + 1))
    im = Image.fromarray(img_u8).crop(box).resize((W, H), Image.BILINEAR)
    a = np.asarray(im, np.float32)
    if p.flip:
        a = a[:, ::-1]
    if p.saturation != 1.0:
        g = a @ np.array([0.299, 0.587, 0.114], np.float32)
        a = g[..., None] + (a - g[..., None]) * p.saturation
    if p.contrast != 1.0:
        a = (a - 127.5) * p.contrast + 127.5
    a = a + p.brightness
    a = np.clip(np.rint(a), 0, 255).astype(np.uint8)
    for (ex1, ey1, ex2, ey2), col in p.erase:
        a[int(ey1 * H):int(math.ceil(ey2 * H)), int(ex1 * W):int(math.ceil(ex2 * W))] = col
    return a


def apply_pixels_device(images, params, out_hw, device):
    """K14: list of uint8 [h,w,3] arrays (host) or uint8 [h,w,3] DEVICE tensors + list[AugParams] -> uint8 torch tensor
    [B,H,W,3] on `device`.  Device tensors (the generator's device-resident image cache) are read where they are: the
    kernel takes one base pointer and a 64-bit offset per image, so the images need not share an allocation."""
    import ctypes as C

    import torch

    from . import _lib
    from .net import Context, _stream_ptr
    ctx = Context.get(device)
    H, Wd = out_hw
    B = len(images)
    arr = (_lib.AugParams * B)()
    resident = B > 0 and all(isinstance(im, torch.Tensor) for im in images)
    off = 0
    chunks = []
    for i, (img, p) in enumerate(zip(images, params)):
        if resident:
            assert img.is_cuda and img.dtype == torch.uint8 and img.is_contiguous() and img.shape[-1] == 3
            a = img
            this_off = img.data_ptr() - images[0].data_ptr()
        else:
            a = np.ascontiguousarray(img[..., :3], np.uint8)
            chunks.append(a.reshape(-1))
            this_off = off
        q = arr[i]
        q.src_offset, q.src_h, q.src_w = this_off, a.shape[0], a.shape[1]
        q.crop_x1, q.crop_y1, q.crop_x2, q.crop_y2 = p.crop
        q.flip, q.brightness, q.contrast, q.saturation = int(p.flip), p.brightness, p.contrast, p.saturation
        q.n_erase = min(3, len(p.erase))
        for e, (rect, col) in enumerate(p.erase[:3]):
            for k in range(4):
                q.erase[e][k] = rect[k]
            for k in range(3):
                q.erase_rgb[e][k] = col[k]
        if not resident:
            off += (a.size + 15) // 16 * 16
    dev = torch.device(device)
    if resident:
        src = images[0]
    else:
        packed = np.zeros(off, np.uint8)
        o = 0
        for c in chunks:
            packed[o:o + c.size] = c
            o += (c.size + 15) // 16 * 16
        src = torch.from_numpy(packed).to(dev)
    prm = torch.from_numpy(np.frombuffer(bytes(arr), np.uint8).copy()).to(dev)
    out = torch.empty((B, H, Wd, 3), dtype=torch.uint8, device=dev)
    _lib.check(ctx.lib.od_augment_batch(ctx.handle, src.data_ptr(), prm.data_ptr(), out.data_ptr(), B, H, Wd,
                                        _stream_ptr()), "od_augment_batch")
    return out


class Generator:
    def __init__(self, input_size, preprocess_input=None, encode_truth=None, random_erasing=True, device=None, workers=None,
                 on_device=False, device_cache=False):
        # device_cache (needs device=): every image is decoded and uploaded ONCE and stays in HBM as a uint8 tensor; from the
        # second epoch on a batch costs the host only its augmentation parameters.  VOC07+12 trainval decoded is ~9 GB --
        # a few per cent of one MI355X's 288 GB -- so the dataset lives where the augmentation kernel reads it.
        self.device_cache = bool(device_cache)
        if self.device_cache and device is None:
            raise ValueError("device_cache=True needs device=")
        # on_device (needs device=): X_batch stays a uint8 DEVICE tensor [B,H,W,3] -- what Trainer.step consumes -- instead of
        # coming back to the host as numpy (the reference's generator feeds Keras from the host; the visual checkers
        # check_generator.py / check_assign.py keep on_device=False and get numpy, as they index pixels on the host).
        # With encode_truth = od.pb.encode_truth_device the targets stay on the device as well: no host round trip per step.
        self.on_device = bool(on_device)
        if self.on_device and device is None:
            raise ValueError("on_device=True needs device=")
        self.input_size = tuple(int(v) for v in input_size)
        self.preprocess_input = preprocess_input
        self.encode_truth = encode_truth
        self.random_erasing = random_erasing
        self.device = device  # e.g. "cuda:0": pixel work runs in od_augment_batch (K14); None: host path (PIL)
        # image decode (and the host augmentation) of a batch runs on a thread pool, the NEXT batch's decode is already in
        # flight while this one is consumed; augmentation parameters are still drawn in index order on the calling thread,
        # so the stream of batches is identical for any worker count
        import os
        self.workers = int(workers if workers is not None else os.environ.get("OD_GEN_WORKERS", min(8, os.cpu_count() or 1)))

    def _load(self, x):
        if isinstance(x, np.ndarray):
            return x[..., :3].astype(np.uint8)
        from .tk.ndimage import load
        return load(x)

    def generate(self, x, ann: ObjectsAnnotation, rng, data_augmentation):
        img = self._load(x)
        p = sample_params(rng, ann, self.random_erasing) if data_augmentation else AugParams()
        out = apply_pixels_host(img, p, self.input_size)
        new_ann = ObjectsAnnotation(ann.path, self.input_size[1], self.input_size[0], ann.classes,
                                    transform_boxes(ann.bboxes, p), ann.difficults)
        return out, new_ann

    def flow(self, X, y, batch_size=16, data_augmentation=False, shuffle=False, seed=0, prefetch=0):
        """-> (infinite iterator of (X_batch [B,H,W,3], y_batch), steps_per_epoch)   (check_generator.py:18)
        prefetch = N > 0 (needs on_device=True): the batches are produced by a background thread, N ahead, on a HIP stream of
        its own -- parameter sampling, packing, the upload, od_augment_batch and encode_truth overlap the training step that
        consumes the previous batch (13.7 -> 10.9 ms per step of scripts/train.py at 32 x 320^2); the stream of batches is the
        same as without prefetch."""
        n = len(X)
        steps = max(1, math.ceil(n / batch_size))
        if prefetch and not self.on_device:
            raise ValueError("prefetch needs on_device=True (device batches on the generator's own stream)")

        def it():
            from concurrent.futures import ThreadPoolExecutor
            rng = np.random.default_rng(seed)
            pool = ThreadPoolExecutor(max_workers=self.workers) if self.workers > 1 else None
            resident = {} if self.device_cache else None  # image index -> uint8 device tensor [h,w,3]

            class _Done:  # a cached image needs no decode: stands in for the pool's future
                def __init__(self, v):
                    self.v = v

                def result(self):
                    return self.v

            def fetch(i):
                i = int(i)
                if resident is not None and i in resident:
                    return _Done(resident[i])
                return pool.submit(self._load, X[i]) if pool is not None else _Done(self._load(X[i]))

            def load_batch(idx):
                return [fetch(i) for i in idx] if (pool is not None or resident is not None) else None

            def to_resident(idx, raw):
                import torch
                out = []
                for i, im in zip(idx, raw):
                    if not isinstance(im, torch.Tensor):
                        im = torch.from_numpy(np.ascontiguousarray(im[..., :3], np.uint8)).to(torch.device(self.device))
                        resident[int(i)] = im
                    out.append(im)
                return out

            while True:
                order = rng.permutation(n) if shuffle else np.arange(n)
                batches = [order[s:s + batch_size] for s in range(0, n, batch_size)]
                nxt = load_batch(batches[0])
                for bi, idx in enumerate(batches):
                    futs, nxt = nxt, (load_batch(batches[bi + 1]) if bi + 1 < len(batches) else None)
                    raw = [f.result() for f in futs] if futs is not None else [self._load(X[i]) for i in idx]
                    if resident is not None:
                        raw = to_resident(idx, raw)
                    prm = [sample_params(rng, y[i], self.random_erasing) if data_augmentation else AugParams() for i in idx]
                    if self.device is not None:
                        xb = apply_pixels_device(raw, prm, self.input_size, self.device)
                        if not self.on_device:
                            xb = xb.cpu().numpy()
                    elif pool is not None:
                        xb = np.stack(list(pool.map(lambda ip: apply_pixels_host(ip[0], ip[1], self.input_size), zip(raw, prm))))
                    else:
                        xb = np.stack([apply_pixels_host(img, p, self.input_size) for img, p in zip(raw, prm)])
                    anns = [ObjectsAnnotation(y[i].path, self.input_size[1], self.input_size[0], y[i].classes,
                                              transform_boxes(y[i].bboxes, p), y[i].difficults)
                            for i, p in zip(idx, prm)]
                    if self.preprocess_input is not None:
                        xb = self.preprocess_input(xb)
                    yb = self.encode_truth(list(anns)) if self.encode_truth is not None else list(anns)
                    yield xb, yb
        if prefetch:
            return _prefetched(it(), int(prefetch), self.device), steps
        return it(), steps


def _prefetched(iterator, depth, device):
    """Run `iterator` in a daemon thread on its own HIP stream, `depth` batches ahead.  Every batch travels with an event
    recorded behind its last kernel; the consumer's current stream waits for it (no host wait), and the tensors are
    registered with that stream so that the caching allocator does not hand their memory back while they are in use."""
    import queue
    import threading

    import torch
    dev = torch.device(device)
    q = queue.Queue(maxsize=depth)
    stream = torch.cuda.Stream(device=dev)
    stop = threading.Event()

    def work():
        try:
            with torch.cuda.device(dev), torch.cuda.stream(stream):
                for item in iterator:
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    while not stop.is_set():
                        try:
                            q.put((item, ev), timeout=0.2)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
        except BaseException as e:  # noqa: BLE001 -- hand the failure to the consumer
            q.put((e, None))

    t = threading.Thread(target=work, daemon=True, name="od_gen-prefetch")
    t.start()

    def gen():
        try:
            while True:
                item, ev = q.get()
                if ev is None:
                    raise item
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                for v in item:
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(cur)
                yield item
        finally:  # consumer closed / garbage-collected the iterator: stop the thread BEFORE the interpreter can tear down
            stop.set()
            try:
                while True:
                    q.get_nowait()
            except queue.Empty:
                pass
            t.join(timeout=10.0)
    return gen()


def create_generator(input_size, preprocess_input=None, encode_truth=None, **kw):
    return Generator(input_size, preprocess_input, encode_truth, **kw)
