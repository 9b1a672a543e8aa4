"""`od.pb`: prior boxes with device-side encode_truth / decode_locs (reference check_assign.py:21,25-27).

`decode_locs(locs, xp=np)` and `encode_truth(y)` keep the reference call shapes; both run the HIP kernels
(od_decode_locs / od_assign_anchors) and hand numpy arrays back, exactly what check_assign.py consumes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, priors as PR
from .net import Context, _stream_ptr

POS_THR, NEG_THR, LOC_SCALE = 0.5, 0.4, 0.1
GMAX = 128


class ObjectsAnnotation:
    """Ground truth of one image (what the generator yields when encode_truth is None: `.classes`, `.bboxes`,
    reference check_generator.py:21).  bboxes are corner form, normalised [0,1]."""

    def __init__(self, path=None, width=0, height=0, classes=(), bboxes=(), difficults=None):
        self.path = path
        self.width, self.height = int(width), int(height)
        self.classes = np.asarray(classes, np.int32).reshape(-1)
        self.bboxes = np.asarray(bboxes, np.float32).reshape(-1, 4)
        self.difficults = (np.zeros(len(self.classes), bool) if difficults is None
                           else np.asarray(difficults, bool).reshape(-1))

    @property
    def num_objects(self):
        return len(self.classes)


class PriorBoxes:
    def __init__(self, input_size=(320, 320), num_classes=20, prior_wh=PR.DEFAULT_PRIOR_WH, device="cuda:0",
                 pos_thr=POS_THR, neg_thr=NEG_THR, loc_scale=LOC_SCALE):
        self.input_size = tuple(int(v) for v in input_size)
        self.num_classes = int(num_classes)
        self.prior_wh = np.asarray(prior_wh, np.float64)
        self.pb_locs = PR.make_priors(self.input_size, self.prior_wh)  # f32 [P,4]
        self.pos_thr, self.neg_thr, self.loc_scale = float(pos_thr), float(neg_thr), float(loc_scale)
        self.device = torch.device(device)
        self._ctx = None
        self._priors_dev = None

    def __len__(self):
        return len(self.pb_locs)

    # -- device plumbing ------------------------------------------------------------------------------------
    def _ensure(self):
        if self._ctx is None:
            self._ctx = Context.get(self.device)
            self._priors_dev = torch.from_numpy(self.pb_locs).to(self.device)
        return self._ctx

    @property
    def priors_device(self):
        self._ensure()
        return self._priors_dev

    def decode_locs(self, locs, xp=np):
        """corner-form offsets [P,4] / [N,P,4] -> boxes; zeros decode to the prior boxes (check_assign.py:27)."""
        from . import ops
        self._ensure()
        if isinstance(locs, torch.Tensor):
            t = locs.to(self.device, torch.float32)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(locs, np.float32))).to(self.device)
        out = ops.decode_locs(t, self._priors_dev, self.loc_scale, clip=False)
        if xp is np:
            return out.cpu().numpy()
        return out

    def encode_batch(self, annotations, return_device=False):
        """list[ObjectsAnnotation] -> y f32 [B,P,2+NC+4] (+ npos [B]) through od_assign_anchors."""
        ctx = self._ensure()
        B, P, NC = len(annotations), len(self.pb_locs), self.num_classes
        gmax = max(1, max((a.num_objects for a in annotations), default=1))
        if gmax > GMAX:
            raise ValueError(f"more than {GMAX} objects in one image")
        gb = np.zeros((B, gmax, 4), np.float32)
        gc = np.zeros((B, gmax), np.int32)
        gn = np.zeros((B,), np.int32)
        for i, a in enumerate(annotations):
            n = a.num_objects
            gb[i, :n], gc[i, :n], gn[i] = a.bboxes, a.classes, n
        dev = self.device
        gb_t, gc_t, gn_t = (torch.from_numpy(v).to(dev) for v in (gb, gc, gn))
        y = torch.empty((B, P, NC + 6), dtype=torch.float32, device=dev)
        npos = torch.empty((B,), dtype=torch.int32, device=dev)
        assigned = torch.empty((B, P), dtype=torch.int32, device=dev)
        wsb = ctx.lib.od_assign_workspace_bytes(B, P, gmax)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        _lib.check(ctx.lib.od_assign_anchors(ctx.handle, self._priors_dev.data_ptr(), gb_t.data_ptr(), gc_t.data_ptr(),
                                             gn_t.data_ptr(), B, P, gmax, NC, self.pos_thr, self.neg_thr,
                                             self.loc_scale, y.data_ptr(), assigned.data_ptr(), npos.data_ptr(),
                                             ws.data_ptr(), wsb, _stream_ptr()), "od_assign_anchors")
        if return_device:
            return y, npos, assigned
        return y.cpu().numpy(), npos.cpu().numpy(), assigned.cpu().numpy()

    def encode_truth_device(self, y):
        """encode_truth whose result stays on the device: list of annotations -> f32 DEVICE tensor [B,P,C] (the generator
        callback for training loops: `Trainer.step(x, y_target=...)` takes it as is)."""
        if isinstance(y, ObjectsAnnotation):
            y = [y]
        return self.encode_batch(list(y), return_device=True)[0]

    def encode_truth(self, y):
        """Generator callback (check_assign.py:21): list of annotations -> array [B,P,C]; one annotation -> [P,C]."""
        if isinstance(y, ObjectsAnnotation):
            return self.encode_batch([y])[0][0]
        return self.encode_batch(list(y))[0]
