"""Device post-processing of the prediction tensor: confidence + decode (K5/K6), exact top-K (K7), NMS (K8).

Replaces the in-graph decode + NMS of reference `ObjectDetector.predict` (voc_validate.py:27;
docs/MODEL.md:54-58 confidence = objectness x class probability, :78-82 same-class NMS).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .net import Context, _stream_ptr

# [BUILD-DEFINED] defaults (the reference does not pin them)
IOU_THRESHOLD = 0.45
PRE_NMS_TOPK = 1024
MAX_DETECTIONS = 200
LOC_SCALE = 0.1


class Postprocessor:
    def __init__(self, batch_size, num_priors, num_classes, priors, device="cuda:0", topk=PRE_NMS_TOPK,
                 max_det=MAX_DETECTIONS, iou_threshold=IOU_THRESHOLD, strict_nms=False, loc_scale=LOC_SCALE):
        self.ctx = Context.get(device)
        self.lib = self.ctx.lib
        dev = torch.device(device)
        self.B, self.P, self.NC = int(batch_size), int(num_priors), int(num_classes)
        self.K, self.max_det = int(topk), int(max_det)
        self.iou_threshold, self.strict, self.loc_scale = float(iou_threshold), int(bool(strict_nms)), float(loc_scale)
        self.priors = torch.as_tensor(priors, dtype=torch.float32).contiguous().to(dev)
        assert self.priors.shape == (self.P, 4)
        B, P, NC, K = self.B, self.P, self.NC, self.K
        # the product path (run -> od_detect) never materialises the confidence tensor; `conf` is filled on demand from the
        # last pred (same kernel code, same bits) for callers that want the dense [B,P,NC] view (tests, head())
        self._conf = None
        self._pred = None
        self._conf_valid = False
        self.fused = P % 2 == 0 and NC <= 76
        self.boxes = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
        self.keys = torch.empty((B, K), dtype=torch.int64, device=dev)  # u64 payload
        self.counts = torch.empty((B,), dtype=torch.int32, device=dev)
        self.keep_flat = torch.empty((B, self.max_det), dtype=torch.int32, device=dev)
        self.keep_count = torch.empty((B,), dtype=torch.int32, device=dev)
        self.ws_topk_bytes = self.lib.od_topk_workspace_bytes(B, P * NC, K)
        self.ws_nms_bytes = self.lib.od_nms_workspace_bytes(B, K)
        # kept detections of the batch as one record block + its pinned host mirror: ONE device->host copy per batch
        self.det = torch.empty((B, 1 + 6 * self.max_det), dtype=torch.float32, device=dev)
        self.det_host = torch.empty((B, 1 + 6 * self.max_det), dtype=torch.float32).pin_memory()
        self._ws_topk = None  # only the three-call path (head / topk / nms separately) needs it
        self.ws_nms = torch.empty((self.ws_nms_bytes,), dtype=torch.uint8, device=dev)
        self.ws_det_bytes = self.lib.od_detect_workspace_bytes(B, P, NC, K)
        self.ws_det = torch.empty((self.ws_det_bytes,), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.od_detect_workspace_init(self.ctx.handle, self.ws_det.data_ptr(), self.ws_det_bytes, B, P, NC,
                                                     _stream_ptr()), "od_detect_workspace_init")
        torch.cuda.current_stream(dev).synchronize()  # the pipelines use this workspace on their own streams

    @property
    def ws_topk(self):
        if self._ws_topk is None:
            self._ws_topk = torch.empty((self.ws_topk_bytes,), dtype=torch.uint8, device=self.priors.device)
        return self._ws_topk

    @property
    def conf(self):
        """Dense confidences f32 [B,P,NC] of the last batch: written by head(), or computed on demand from the pred that
        run() saw (od_head_postprocess: the same confidence code as the fused path, bit-identical values)."""
        if self._conf is None:
            self._conf = torch.empty((self.B, self.P, self.NC), dtype=torch.float32, device=self.priors.device)
        if not self._conf_valid:
            if self._pred is None:
                raise _lib.OdError("Postprocessor.conf: nothing has been processed yet")
            self.head(self._pred)
        return self._conf

    def head(self, pred: torch.Tensor, clip=True):
        assert pred.dtype == torch.float32 and pred.is_contiguous() and tuple(pred.shape) == (self.B, self.P, self.NC + 6)
        if self._conf is None:
            self._conf = torch.empty((self.B, self.P, self.NC), dtype=torch.float32, device=self.priors.device)
        _lib.check(self.lib.od_head_postprocess(self.ctx.handle, pred.data_ptr(), self.priors.data_ptr(),
                                                self._conf.data_ptr(), self.boxes.data_ptr(), self.B, self.P, self.NC,
                                                self.loc_scale, int(clip), _stream_ptr()), "od_head_postprocess")
        self._pred, self._conf_valid = pred, True
        return self._conf, self.boxes

    def topk(self, conf: torch.Tensor, conf_threshold: float):
        assert conf.dtype == torch.float32 and conf.is_contiguous()
        _lib.check(self.lib.od_topk_scores(self.ctx.handle, conf.data_ptr(), self.B, self.P * self.NC, self.K,
                                           float(conf_threshold), self.keys.data_ptr(), self.counts.data_ptr(),
                                           self.ws_topk.data_ptr(), self.ws_topk_bytes, _stream_ptr()), "od_topk_scores")
        return self.keys, self.counts

    def nms(self, boxes: torch.Tensor, keys: torch.Tensor, counts: torch.Tensor):
        _lib.check(self.lib.od_nms(self.ctx.handle, boxes.data_ptr(), keys.data_ptr(), counts.data_ptr(), self.B,
                                   self.P, self.NC, self.K, self.iou_threshold, self.strict, self.max_det,
                                   self.keep_flat.data_ptr(), self.keep_count.data_ptr(), self.ws_nms.data_ptr(),
                                   self.ws_nms_bytes, _stream_ptr()), "od_nms")
        return self.keep_flat, self.keep_count

    def gather(self):
        """Queue the gather of this batch's kept detections (od_gather_detections) and its copy into the pinned host block
        on the current stream; read with detections_host() after the stream / event has been waited for."""
        _lib.check(self.lib.od_gather_detections_pred(self.ctx.handle, self._pred.data_ptr(), self.boxes.data_ptr(),
                                                      self.keep_flat.data_ptr(), self.keep_count.data_ptr(), self.B, self.P,
                                                      self.NC, self.max_det, self.det.data_ptr(), _stream_ptr()),
                   "od_gather_detections_pred")
        self.det_host.copy_(self.det, non_blocking=True)

    def detections_host(self, n_valid):
        """-> [(flat i32 [n], conf f32 [n], boxes f32 [n,4])] for the first n_valid images, from the pinned block."""
        import numpy as np
        a = self.det_host.numpy()
        out = []
        for b in range(n_valid):
            n = int(a[b, :1].view(np.int32)[0])
            rows = a[b, 1:1 + 6 * n].reshape(n, 6)
            out.append((rows[:, 0].copy().view(np.int32), rows[:, 1].copy(), rows[:, 2:6].copy()))
        return out

    def run(self, pred: torch.Tensor, conf_threshold: float):
        """pred [B,P,C] -> (keep_flat i32 [B,max_det] (-1 padded), keep_count i32 [B]); boxes / keys stay on device.
        ONE C-ABI call (od_detect, five launches); run_unfused() is the same result through head -> topk -> nms."""
        if not self.fused:
            return self.run_unfused(pred, conf_threshold)
        assert pred.dtype == torch.float32 and pred.is_contiguous() and tuple(pred.shape) == (self.B, self.P, self.NC + 6)
        _lib.check(self.lib.od_detect(self.ctx.handle, pred.data_ptr(), self.priors.data_ptr(), self.B, self.P, self.NC,
                                      self.loc_scale, 1, float(conf_threshold), self.K, self.iou_threshold, self.strict,
                                      self.max_det, self.boxes.data_ptr(), None, self.keys.data_ptr(), self.counts.data_ptr(),
                                      self.keep_flat.data_ptr(), self.keep_count.data_ptr(), self.ws_det.data_ptr(),
                                      self.ws_det_bytes, self.ws_nms.data_ptr(), self.ws_nms_bytes, _stream_ptr()), "od_detect")
        self._pred, self._conf_valid = pred, False
        return self.keep_flat, self.keep_count

    def run_unfused(self, pred: torch.Tensor, conf_threshold: float):
        """The three-call path (od_head_postprocess -> od_topk_scores -> od_nms): materialises conf, same kept indices."""
        conf, boxes = self.head(pred)
        keys, counts = self.topk(conf, conf_threshold)
        return self.nms(boxes, keys, counts)
