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
        self.conf = torch.empty((B, P, NC), dtype=torch.float32, device=dev)
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
        self.ws_topk = torch.empty((self.ws_topk_bytes,), dtype=torch.uint8, device=dev)
        self.ws_nms = torch.empty((self.ws_nms_bytes,), dtype=torch.uint8, device=dev)

    def head(self, pred: torch.Tensor, clip=True):
        assert pred.dtype == torch.float32 and pred.is_contiguous() and tuple(pred.shape) == (self.B, self.P, self.NC + 6)
        _lib.check(self.lib.od_head_postprocess(self.ctx.handle, pred.data_ptr(), self.priors.data_ptr(),
                                                self.conf.data_ptr(), self.boxes.data_ptr(), self.B, self.P, self.NC,
                                                self.loc_scale, int(clip), _stream_ptr()), "od_head_postprocess")
        return self.conf, self.boxes

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
        _lib.check(self.lib.od_gather_detections(self.ctx.handle, self.conf.data_ptr(), self.boxes.data_ptr(),
                                                 self.keep_flat.data_ptr(), self.keep_count.data_ptr(), self.B, self.P,
                                                 self.NC, self.max_det, self.det.data_ptr(), _stream_ptr()),
                   "od_gather_detections")
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
        """pred [B,P,C] -> (keep_flat i32 [B,max_det] (-1 padded), keep_count i32 [B]); conf/boxes stay on device."""
        conf, boxes = self.head(pred)
        keys, counts = self.topk(conf, conf_threshold)
        return self.nms(boxes, keys, counts)
