"""`tk.dl.od.ObjectDetector`: the drop-in boundary of the hot path.

Keeps the call surface the reference scripts use (voc_validate.py:25-27, voc_evaluate.py:25-27, check_assign.py:19):
    ObjectDetector.load_voc(batch_size, input_size=(320,320), keep_aspect=False, strict_nms=False, use_multi_gpu=True)
    od.predict(X, conf_threshold=...) -> list of per-image predictions (.classes, .confs, .bboxes, .plot(x, names))
    od.pb.encode_truth / od.pb.decode_locs
and runs everything below it on MI355X through libodhip.so.  use_multi_gpu shards images by index over the ranks of
an already-initialised torch.distributed job (one process per GPU); inference needs no collective on the data path,
results are gathered on the host.
"""
from __future__ import annotations

import os
import pathlib

import numpy as np
import torch

from . import weights as W
from .net import Net
from .pb import ObjectsAnnotation, PriorBoxes
from .postprocess import Postprocessor

DEFAULT_CONF_THRESHOLD = 0.01  # [BUILD-DEFINED]: mAP needs the low-confidence tail (voc_evaluate.py:27 passes 0.6)
VOC_WEIGHTS_ENV = "OD_VOC_WEIGHTS"


class ObjectsPrediction:
    """Detections of one image: classes i32 [n], confs f32 [n], bboxes f32 [n,4] (corner form, normalised [0,1] of the
    ORIGINAL image: keep_aspect=False is a plain resize, so normalised coordinates are resize-invariant)."""

    def __init__(self, classes, confs, bboxes, flat_indices=None):
        self.classes = np.asarray(classes, np.int32)
        self.confs = np.asarray(confs, np.float32)
        self.bboxes = np.asarray(bboxes, np.float32).reshape(-1, 4)
        self.flat_indices = None if flat_indices is None else np.asarray(flat_indices, np.int32)

    def plot(self, x, class_names=None):
        from .tk import ml
        return ml.plot_objects(x, self.classes, self.confs, self.bboxes, class_names)

    def __len__(self):
        return len(self.classes)


from .imageio import decode_chunk_into_shm, load_image  # noqa: E402,F401  (host-side decode + resize; re-exported)


def dist_info(use_multi_gpu=True):
    """(rank, world) of the torch.distributed job this process belongs to (one process per GPU), else (0, 1)."""
    if use_multi_gpu and torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank(), torch.distributed.get_world_size()
    return 0, 1


def shard_indices(n, rank, world):
    """Images are independent: rank r takes indices r, r+world, ... -- no collective on the data path."""
    return list(range(rank, n, world))


def gather_results(local: dict, world: int) -> dict:
    """Host-side gather of the per-rank {index: prediction} maps (variable-length results; every rank gets all)."""
    if world <= 1:
        return local
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, local)
    out = {}
    for d in gathered:
        out.update(d)
    return out


_QUEUE_SETS = {}  # (device index, n) -> [torch.cuda.Stream] that were measured to run concurrently


def _streams_overlap(a, b, spin_cycles):
    """True when a kernel queued on `b` finishes while an earlier, long kernel on `a` is still running (different hardware
    queues); False when `b`'s kernel had to wait for it (same queue)."""
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
        ea.record()
    with torch.cuda.stream(b):
        torch.cuda._sleep(1)
        eb.record()
    eb.synchronize()
    free = not ea.query()
    ea.synchronize()
    return free


def stream_queue_sets(device, n, candidates=6, seed_streams=(), beside=None):
    """n streams of `device` that overlap pairwise -- and with the stream `beside`, if given (the trainer's side streams must
    run next to its main stream) -- chosen greedily from `candidates` fresh streams by the probe above (about 2 ms per pair,
    a few dozen ms once per process); cached.  Falls back to creation order when the probe cannot tell streams apart (e.g.
    a runtime that serialises everything)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), n,
           None if beside is None else beside.cuda_stream)
    if key in _QUEUE_SETS:
        return _QUEUE_SETS[key]
    if not hasattr(torch.cuda, "_sleep"):  # no spin kernel to probe with: creation order (what rounds 1-2 started from)
        _QUEUE_SETS[key] = list(seed_streams)[:n] + [torch.cuda.Stream(device=device) for _ in range(n - min(n, len(seed_streams)))]
        return _QUEUE_SETS[key]
    with torch.cuda.device(device):
        cands = list(seed_streams) + [torch.cuda.Stream(device=device) for _ in range(max(candidates, n) - len(seed_streams))]
        for c in cands:  # every stream's first launch (queue creation) happens outside the probes
            with torch.cuda.stream(c):
                torch.cuda._sleep(1000)
        torch.cuda.synchronize(device)
        # spin length: ~2 ms of a kernel that does nothing but read the clock (calibrated, the tick rate is not assumed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(cands[0]):
            torch.cuda._sleep(1000)  # first launch: module load
            e0.record()
            torch.cuda._sleep(200000)
            e1.record()
        e1.synchronize()
        per_ms = 200000 / max(e0.elapsed_time(e1), 1e-3)
        spin = int(max(1000, 2.0 * per_ms))
        chosen = []
        fixed = [] if beside is None else [beside]
        for c in cands:
            if len(chosen) == n:
                break
            if all(_streams_overlap(q, c, spin) and _streams_overlap(c, q, spin) for q in fixed + chosen):
                chosen.append(c)
        for c in cands:  # not enough distinguishable queues: take the rest in creation order
            if len(chosen) == n:
                break
            if c not in chosen:
                chosen.append(c)
    _QUEUE_SETS[key] = chosen
    return chosen


class _Pipeline:
    """One batch in flight: its own activation buffers (Net), post-processing buffers and HIP stream."""

    def __init__(self, net, post, stream):
        self.net, self.post, self.stream = net, post, stream
        self.done = torch.cuda.Event()


class ObjectDetector:
    def __init__(self, params, batch_size=16, input_size=(320, 320), keep_aspect=False, strict_nms=False,
                 use_multi_gpu=True, device=None, prior_wh=None, n_inflight=None, precision=None):
        if device is None:  # one process per GPU; the modulo only matters when several ranks rehearse on one GPU
            device = f"cuda:{int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count())}"
        if not torch.cuda.is_available():
            from ._lib import OdError
            raise OdError("ObjectDetector needs an MI355X (no GPU visible); there is no CPU path")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.batch_size = int(batch_size)
        self.input_size = tuple(int(v) for v in input_size)
        self.keep_aspect, self.strict_nms, self.use_multi_gpu = bool(keep_aspect), bool(strict_nms), bool(use_multi_gpu)
        self.params = params
        self.num_classes, _, _ = W.infer_arch(params)
        kw = {} if prior_wh is None else {"prior_wh": prior_wh}
        self.pb = PriorBoxes(self.input_size, self.num_classes, device=self.device, **kw)
        n = int(n_inflight) if n_inflight is not None else int(os.environ.get("OD_INFLIGHT", 3))  # explicit argument wins
        # precision: "f16" (default, the throughput plan) or "mixed" (f32 residual stream + split operands in the last layers:
        # north_star's 1e-3 logit tolerance at any logit scale, net.py); None = $OD_PRECISION or "f16"
        self.net = Net(params, self.batch_size, self.input_size, device=self.device, overlapped=n > 1, precision=precision)
        self.precision = self.net.precision
        assert self.net.P == len(self.pb)
        self.post = Postprocessor(self.batch_size, self.net.P, self.num_classes, self.pb.pb_locs, device=self.device,
                                  strict_nms=self.strict_nms, loc_scale=self.pb.loc_scale)
        # Batches in flight (submit / collect): every pipeline has its own buffers and stream, so the launch ramps, tile
        # tails, epilogue write bursts and the small post-processing kernels of one batch overlap the convolutions of the
        # next (measured +23 % images/s at batch 32 with 2-3 in flight, profiles/r01/inflight_sweep.txt).  Pipeline 0 =
        # (self.net, self.post) on the caller's stream is what predict_batch_device uses.
        self._pipes = [_Pipeline(self.net, self.post, torch.cuda.Stream(device=self.device))]
        for _ in range(max(1, n) - 1):
            net = Net(params, self.batch_size, self.input_size, device=self.device, overlapped=True,
                      share_weights_with=self.net, precision=self.precision, stream_stages=self.net.stream_stages,
                      split=self.net.split, wide_fpn=self.net.wide_fpn)
            post = Postprocessor(self.batch_size, net.P, self.num_classes, self.pb.pb_locs, device=self.device,
                                 strict_nms=self.strict_nms, loc_scale=self.pb.loc_scale)
            self._pipes.append(_Pipeline(net, post, torch.cuda.Stream(device=self.device)))
        self._next = 0
        self._calibrated = len(self._pipes) < 2 or os.environ.get("OD_INFLIGHT_CALIBRATE", "1") == "0"

    # -- construction -----------------------------------------------------------------------------------------
    @classmethod
    def load_voc(cls, batch_size, input_size=(320, 320), keep_aspect=False, strict_nms=False, use_multi_gpu=True,
                 weights=None, **kw):
        """VOC-trained detector.  The reference pulls an LFS `*.h5` (.gitattributes:5); offline this build reads its own
        .npz from `weights`, $OD_VOC_WEIGHTS, or ./weights/voc07+12_<H>x<W>.npz -- a LOCAL path, never a download."""
        path = weights or os.environ.get(VOC_WEIGHTS_ENV) or \
            pathlib.Path("weights") / f"voc07+12_{input_size[0]}x{input_size[1]}.npz"
        path = pathlib.Path(path)
        if not path.exists():
            raise FileNotFoundError(
                f"VOC weights not found at {path}. Trained weights are not distributable offline; pass weights=<npz>, "
                f"set ${VOC_WEIGHTS_ENV}, or build a synthetic detector with ObjectDetector.synthetic(...).")
        params, meta = W.load(path)
        if "prior_wh" in meta:
            kw.setdefault("prior_wh", meta["prior_wh"])
        return cls(params, batch_size, input_size, keep_aspect, strict_nms, use_multi_gpu, **kw)

    @classmethod
    def synthetic(cls, batch_size, input_size=(320, 320), seed=2, num_classes=20, **kw):
        """Random-init detector of the VOC architecture (benchmarks / parity tests)."""
        return cls(W.random_init(seed, num_classes), batch_size, input_size, **kw)

    # -- inference --------------------------------------------------------------------------------------------
    def predict_batch_device(self, x_u8: torch.Tensor, conf_threshold=DEFAULT_CONF_THRESHOLD, graph=False):
        """uint8 [B,H,W,3] on device -> (keep_flat [B,max_det], keep_count [B]) on device.  The timed hot path."""
        torch.cuda.current_stream(self.device).wait_stream(self._pipes[0].stream)  # pipeline 0's buffers may be in flight
        pred = self.net.forward(x_u8, graph=graph)
        return self.post.run(pred, conf_threshold)

    # -- batches in flight ------------------------------------------------------------------------------------------
    @property
    def n_inflight(self):
        return len(self._pipes)

    def submit(self, x_u8: torch.Tensor, conf_threshold=DEFAULT_CONF_THRESHOLD, graph=False, gather=False) -> int:
        """Queue one batch on the next pipeline's stream and return its ticket.  Never blocks the host: a pipeline's new
        batch is stream-ordered behind its previous one (whose results it overwrites -- collect() them first)."""
        if not self._calibrated:
            self._pick_streams()
        i = self._next
        self._next = (i + 1) % len(self._pipes)
        p = self._pipes[i]
        p.stream.wait_stream(torch.cuda.current_stream(self.device))  # x_u8 was produced on the caller's stream
        with torch.cuda.stream(p.stream):
            pred = p.net.forward(x_u8, graph=graph)
            p.post.run(pred, conf_threshold)
            if gather:
                p.post.gather()  # kept detections -> one record block -> pinned host memory, still on this stream
            p.done.record()
        x_u8.record_stream(p.stream)
        return i

    def _pick_streams(self):
        """One stream per pipeline, on DIFFERENT hardware queues.  HIP deals streams onto a few hardware queues (4 by
        default) in creation order, and kernels of two streams that share a queue never overlap: with 1-2 foreign streams
        created first the same code ran at 15.9 k instead of 18.8 k images/s.  Instead of timing real steps on every stream
        combination (rounds 1-2), the queue map is MEASURED once per process and device with a probe that does no real work
        (stream_queue_sets) and cached; every detector of the process takes its streams from that set."""
        self._calibrated = True
        sts = stream_queue_sets(self.device, len(self._pipes), seed_streams=[p.stream for p in self._pipes])
        for p, st in zip(self._pipes, sts):
            p.stream = st
        self._next = 0

    def collect(self, ticket: int):
        """Wait for the batch submitted with `ticket`; -> (keep_flat [B,max_det], keep_count [B]) on device."""
        p = self._pipes[ticket]
        p.done.synchronize()
        return p.post.keep_flat, p.post.keep_count

    def synchronize(self):
        for p in self._pipes:
            p.stream.synchronize()

    def _collect(self, n_valid, scales=None, post=None):
        """Per-image predictions of a collected batch from its pinned record block (submit(..., gather=True) queued the
        gather kernel and the single device->host copy behind the NMS; no per-image device work or synchronisation)."""
        post = post or self.post
        NC = self.num_classes
        out = []
        for b, (k, confs, bxs) in enumerate(post.detections_host(n_valid)):
            if scales is not None and scales[b] != (1.0, 1.0):  # keep_aspect: canvas coordinates -> image coordinates
                sx, sy = scales[b]
                bxs = np.clip(bxs / np.array([sx, sy, sx, sy], np.float32), 0.0, 1.0)
            out.append(ObjectsPrediction(k % NC, confs, bxs, k))
        return out

    def _decode_procs(self, n, nbytes):
        """Lazily created pool of spawned decode workers + their shared-memory staging block (kept for the detector's life)."""
        import atexit
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        from multiprocessing import shared_memory
        st = getattr(self, "_dproc", None)
        if st is not None and (st[2] != n or st[1].size < nbytes):
            self.close_decode_pool()
            st = None
        if st is None:
            pool = ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn"))
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
            st = self._dproc = (pool, shm, n)
            atexit.register(self.close_decode_pool)
        return st[0], st[1]

    def close_decode_pool(self):
        st = getattr(self, "_dproc", None)
        self._dproc = None
        if st is not None:
            st[0].shutdown(wait=True, cancel_futures=True)
            st[1].close()
            try:
                st[1].unlink()
            except FileNotFoundError:
                pass

    def predict(self, X, conf_threshold=DEFAULT_CONF_THRESHOLD, verbose=0):
        """X: sequence of image paths or uint8 arrays -> list[ObjectsPrediction] in input order."""
        X = list(X)
        rank, world = dist_info(self.use_multi_gpu)
        mine = shard_indices(len(X), rank, world)
        results = {}
        B = self.batch_size
        pending = []  # (ticket, image indices, scales): decode + upload of batch k+1.. overlaps the device work of batch k

        def drain(entry):
            ticket, idx, scales = entry
            self.collect(ticket)
            for i, pr in zip(idx, self._collect(len(idx), scales, self._pipes[ticket].post)):
                results[i] = pr

        # Host side: JPEG decode + resize is the slow half of a real predict() (a few ms per image against ~50 us of device
        # time), so the images of the next batches are decoded by a pool while this batch runs.  Default: threads (PIL
        # releases the GIL for decode and resize; ~3x on 4+ cores, then GIL-bound) writing straight into rotating pinned
        # staging buffers that are uploaded asynchronously.  OD_DECODE_PROCS=N: N spawned worker PROCESSES (numpy + PIL
        # only, never the GPU) writing into one shared-memory staging block -- scales with the cores (measured 9.5 k
        # decodes/s on 16); opt-in because spawn re-imports the caller's __main__ (needs the usual __main__ guard).
        batches = [mine[s:s + B] for s in range(0, len(mine), B)]
        ahead = 2
        nb = ahead + 2  # a staging buffer is busy from the start of its decode until its upload has completed
        nprocs = int(os.environ.get("OD_DECODE_PROCS", "0"))
        img_bytes = self.input_size[0] * self.input_size[1] * 3
        if nprocs > 0:
            pool, shm = self._decode_procs(nprocs, nb * B * img_bytes)
            stage = np.ndarray((nb, B) + tuple(self.input_size) + (3,), np.uint8, buffer=shm.buf)

            class _Chunk:  # one future = 4 images; .result() of image j picks its scale out of the chunk's list
                def __init__(self, fut, j):
                    self.fut, self.j = fut, j

                def result(self):
                    return self.fut.result()[self.j]

            def start(bi):
                k = bi % nb  # free: its upload was synchronous
                idxs = batches[bi]
                out = []
                for c0 in range(0, len(idxs), 4):
                    part = idxs[c0:c0 + 4]
                    fut = pool.submit(decode_chunk_into_shm, shm.name, [(k * B + c0 + j) * img_bytes for j in range(len(part))],
                                      [X[i] for i in part], tuple(self.input_size), self.keep_aspect)
                    out += [_Chunk(fut, j) for j in range(len(part))]
                return out

            def upload(bi):
                return torch.from_numpy(stage[bi % nb]).to(self.device)
            own_pool = None
        else:
            from concurrent.futures import ThreadPoolExecutor
            nthreads = int(os.environ.get("OD_DECODE_THREADS", "0")) or min(8, os.cpu_count() or 1)
            if getattr(self, "_hbufs", None) is None:
                self._hbufs = [[torch.zeros((B,) + tuple(self.input_size) + (3,), dtype=torch.uint8).pin_memory(), None]
                               for _ in range(nb)]
            own_pool = pool = ThreadPoolExecutor(max_workers=max(1, nthreads))

            def load_into(x, dst):  # worker thread: decode + resize, then the (slow, uncached) write into pinned memory
                img, sc = load_image(x, self.input_size, self.keep_aspect, True)
                np.copyto(dst, img)
                return sc

            def start(bi):
                hb = self._hbufs[bi % nb]
                if hb[1] is not None:
                    hb[1].synchronize()  # the upload that last read this pinned buffer
                    hb[1] = None
                host = hb[0].numpy()
                return [pool.submit(load_into, X[i], host[j]) for j, i in enumerate(batches[bi])]

            def upload(bi):
                hb = self._hbufs[bi % nb]
                x = hb[0].to(self.device, non_blocking=True)
                hb[1] = torch.cuda.Event()
                hb[1].record(torch.cuda.current_stream(self.device))
                return x
        try:
            futs = {bi: start(bi) for bi in range(min(ahead + 1, len(batches)))}
            for bi, idx in enumerate(batches):
                if bi + ahead + 1 < len(batches):
                    futs[bi + ahead + 1] = start(bi + ahead + 1)
                scales = [f.result() for f in futs.pop(bi)]
                x = upload(bi)
                if len(pending) == len(self._pipes):  # the pipeline about to be reused still holds unread results
                    drain(pending.pop(0))
                pending.append((self.submit(x, conf_threshold, gather=True), idx, scales))
        finally:
            if own_pool is not None:
                own_pool.shutdown(wait=True)
        for entry in pending:
            drain(entry)
        results = gather_results(results, world)
        return [results[i] for i in range(len(X))]
