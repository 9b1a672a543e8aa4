"""Training step of the detector on MI355X: forward in BatchNorm-training mode, prior-box assignment, loss,
backward through every conv / BN / activation, data-parallel gradient all-reduce, SGD with per-layer LR multipliers.

reference: the (unseen) `fit` path of tk.dl.od.ObjectDetector -- generator -> encode_truth (check_assign.py:21) ->
losses (docs/MODEL.md:33-52) -> per-layer learning rates (docs/MODEL.md:84-90: shared prediction-module layers x 1/3,
base network x 1/100).  Every tensor op is a libodhip.so kernel; torch owns the HBM buffers and the process group.

Numerics: activations and activation gradients f16 (gradients multiplied by `loss_scale`), MFMA accumulation and all
statistics / parameter gradients / master weights f32.  BatchNorm statistics are per rank [BUILD-DEFINED, SURVEY.md §7];
only gradients are exchanged.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib, train_ops as T, weights as W
from .net import Context, _stream_ptr
from .pb import PriorBoxes

BN_MOMENTUM = 0.99  # Keras default


LR_MULTIPLIERS = {"b.": 0.01, "h.": 1.0 / 3.0}  # docs/MODEL.md:84-90: base network x 1/100, shared layers x 1/(times shared)


def lr_multiplier(name: str, table=None) -> float:
    """docs/MODEL.md:84-90.  `table`: {layer-name prefix: multiplier}, first matching prefix wins, no match = 1."""
    for prefix, m in (LR_MULTIPLIERS if table is None else table).items():
        if name.startswith(prefix):
            return float(m)
    return 1.0


class _Node:
    __slots__ = ("name", "x", "out", "res", "res_mode", "H", "W", "Cin", "Cout", "k", "stride", "act", "bn", "z",
                 "mean", "rstd", "scale", "shift", "first", "pred_off", "pred_rows", "need_dx")


class Trainer:
    def __init__(self, params, batch_size, input_size=(320, 320), device="cuda:0", lr=1e-3, momentum=0.9,
                 weight_decay=0.0, loss_scale=1024.0, box_mode="smooth_l1", backbone_act=("leaky", 0.1),
                 head_act=("elu", 1.0), comm=None, world_size=1, grad_payload=None, dynamic_loss_scale=True,
                 lr_multipliers=None, prior_wh=None):
        self.ctx = Context.get(device)
        self.lib = self.ctx.lib
        self.device = torch.device(device)
        self.B = int(batch_size)
        self.H0, self.W0 = int(input_size[0]), int(input_size[1])
        self.num_classes, self.neck_ch, self.tower = W.infer_arch(params)
        self.C = self.num_classes + 6
        self.lr, self.momentum, self.weight_decay = float(lr), float(momentum), float(weight_decay)
        self.loss_scale, self.box_mode = float(loss_scale), box_mode
        # per-layer learning-rate multipliers (docs/MODEL.md:84-90 by default: the reference fine-tunes a PRE-TRAINED base
        # network at 1/100; a from-scratch run passes {"h.": 1/3} to train the base network at the full rate)
        self.lr_multipliers = dict(LR_MULTIPLIERS if lr_multipliers is None else lr_multipliers)
        self.comm, self.world = comm, int(world_size)
        # gradient all-reduce payload: "f32" (exact sum) or "bf16" (BASELINE.json configs[4]: half the xGMI bytes; each rank's
        # f32 gradients are rounded to bf16, summed by RCCL in bf16, widened back to f32 before the optimizer)
        self.grad_payload = grad_payload or os.environ.get("OD_TRAIN_GRAD_PAYLOAD", "f32")
        if self.grad_payload not in ("f32", "bf16"):
            raise ValueError(f"grad_payload must be 'f32' or 'bf16', got {self.grad_payload!r}")
        # loss-scale guard: a non-finite value in the flat gradient buffer (one f16 overflow in a loss-scaled dz is enough)
        # makes the device skip the update; the host learns about it one step later and halves the scale
        self.dynamic_loss_scale = bool(dynamic_loss_scale)
        self.skipped_steps = 0
        self._good_steps = 0
        self.loss_scale_growth_interval = 2000
        # prior_wh: the [3, 8, 2] table of prior sizes in grid-cell units (priors.fit: KMeans over the training boxes,
        # docs/MODEL.md:29-31); None = the frozen default table
        self.pb = PriorBoxes((self.H0, self.W0), self.num_classes, device=self.device,
                             **({} if prior_wh is None else {"prior_wh": prior_wh}))
        self.P = len(self.pb)
        dev = self.device
        # od_conv_desc.tile_cfg of the forward / backward-data convolutions: -1 = fastest launch on an idle chip, -2 = least
        # CU x time (the backward-data chain shares the chip with the weight-gradient stream)
        self.fwd_tile_cfg = int(os.environ.get("OD_TRAIN_FWD_CFG", "-1"))
        self.bwd_tile_cfg = int(os.environ.get("OD_TRAIN_BWD_CFG", "-1"))

        # ---- flat f32 parameter / gradient / momentum buffers -------------------------------------------------
        self.specs = {s[0]: s for s in W.layer_specs(self.num_classes, self.neck_ch, self.tower)}
        self.seg = {}  # (layer, kind) -> (offset, numel)
        off = 0
        for name, (_n, cin, cout, k, _s, bn) in self.specs.items():
            for kind, n in (("w", cout * k * k * cin),) + ((("gamma", cout), ("beta", cout)) if bn else (("bias", cout),)):
                self.seg[(name, kind)] = (off, n)
                off += (n + 3) // 4 * 4
        self.n_flat = off
        host = np.zeros(off, np.float32)
        for (name, kind), (o, n) in self.seg.items():
            host[o:o + n] = np.asarray(params[f"{name}.{kind}"], np.float32).reshape(-1)
        self.params = torch.from_numpy(host).to(dev)
        self.grads = torch.zeros_like(self.params)
        self.mom = torch.zeros_like(self.params)
        # BatchNorm running statistics: ONE flat buffer (views per layer), so that a skipped step can put back the copy taken
        # before its forward pass with one launch (od_copy_if_nonzero)
        roff, rviews = 0, {}
        for n, sp in self.specs.items():
            if sp[5]:
                rviews[n] = roff
                roff += 2 * ((sp[2] + 3) // 4 * 4)
        rhost = np.zeros(max(roff, 4), np.float32)
        for n, o in rviews.items():
            c = self.specs[n][2]
            cp = (c + 3) // 4 * 4
            rhost[o:o + c] = np.asarray(params[n + ".mean"], np.float32)
            rhost[o + cp:o + cp + c] = np.asarray(params[n + ".var"], np.float32)
        self.run_stats = torch.from_numpy(rhost).to(dev)
        self.run_stats_saved = torch.empty_like(self.run_stats)
        self.run_mean = {n: self.run_stats[o:o + self.specs[n][2]] for n, o in rviews.items()}
        self.run_var = {n: self.run_stats[o + (self.specs[n][2] + 3) // 4 * 4:][:self.specs[n][2]] for n, o in rviews.items()}
        maxpad = 2048
        self.ones = torch.ones(maxpad, dtype=torch.float32, device=dev)
        self.zeros = torch.zeros(maxpad, dtype=torch.float32, device=dev)
        self.in_scale = torch.full((32,), 1.0 / 255.0, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(maxpad, dtype=torch.float32, device=dev)  # sink for unused per-channel outputs
        # ---- f16 packs ------------------------------------------------------------------------------------------
        self.wf, self.wb = {}, {}
        for name, (_n, cin, cout, k, _s, _bn) in self.specs.items():
            if name == "b.conv0":
                self.wf[name] = torch.zeros((32, 32), dtype=torch.float16, device=dev)
                continue
            cp, kp = _lib.conv_weight_dims(cout, cin, k)
            self.wf[name] = torch.zeros((cp, kp), dtype=torch.float16, device=dev)
            cp2, kp2 = _lib.conv_weight_dims(cin, cout, k)
            self.wb[name] = torch.zeros((cp2, kp2), dtype=torch.float16, device=dev)
        self._repack()
        # ---- graph ------------------------------------------------------------------------------------------------
        self.tensors = {"img": torch.zeros((self.B, self.H0, self.W0, 3), dtype=torch.uint8, device=dev)}
        self.gradbuf = {}
        self.nodes = []
        self._build(backbone_act, head_act)
        self.pred = torch.zeros((self.B, self.P, self.C), dtype=torch.float32, device=dev)
        self.grad_pred = torch.zeros_like(self.pred)
        self.losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self.loss_ws = torch.empty(self.lib.od_loss_workspace_bytes(self.B, self.P), dtype=torch.uint8, device=dev)
        mx = max(self.B * (n.H // n.stride) * (n.W // n.stride) * n.Cout for n in self.nodes)
        # dz of the node being processed: three rotating buffers, because the weight gradient of node k runs on a second
        # stream while the main stream already computes node k-1's dz (backward()); OD_TRAIN_WSTREAM=0 -> one buffer, one stream
        self.use_wstream = os.environ.get("OD_TRAIN_WSTREAM", "1") != "0"
        self.dzs = [torch.empty(mx, dtype=torch.float16, device=dev) for _ in range(3 if self.use_wstream else 1)]
        self.dz = self.dzs[0]
        # the side streams (weight gradients, collectives) must sit on other hardware queues than the main stream, or nothing
        # overlaps: which queue a fresh stream lands on depends on how many streams the process created before, so they are
        # picked with the measured queue-overlap probe of detector.stream_queue_sets (12.9 instead of 10.6 ms per step when
        # the trainer was built after a few detectors in one process)
        from .detector import stream_queue_sets
        with torch.cuda.device(dev):
            side = stream_queue_sets(dev, 2, beside=torch.cuda.current_stream(dev)) if self.use_wstream else [None, None]
        self._side_streams = side
        self.wstream = side[0] if self.use_wstream else None
        # OD_TRAIN_WSTREAM_CUS="first:count[:stride]": the weight-gradient stream confined to `count` CUs starting at `first`
        # (every `stride`-th CU; an experiment knob -- profiles/r03/train_cu_mask.txt)
        spec = os.environ.get("OD_TRAIN_WSTREAM_CUS", "")
        if self.use_wstream and spec:
            parts = [int(v) for v in spec.split(":")]
            first, count, stride = parts[0], parts[1], (parts[2] if len(parts) > 2 else 1)
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            bits = [0] * ((ncu + 31) // 32)
            for i in range(count):
                cu = (first + i * stride) % ncu
                bits[cu // 32] |= 1 << (cu % 32)
            arr = (C.c_uint32 * len(bits))(*bits)
            hnd = C.c_void_p()
            _lib.check(self.lib.od_stream_create_cu_mask(self.ctx.handle, arr, len(bits), C.byref(hnd)), "od_stream_create_cu_mask")
            self._masked_stream = hnd
            self.wstream = torch.cuda.ExternalStream(hnd.value, device=dev)
        self._wg_done = [None] * len(self.dzs)
        # gradient exchange: buckets of whole layers from the END of the flat buffer (the order backward finishes them),
        # each all-reduced on its own stream as soon as its last layer is final -> the RCCL traffic overlaps the rest of
        # the backward pass.  OD_TRAIN_BUCKET_MB=0 -> one all-reduce after backward.
        self.bucket_mb = float(os.environ.get("OD_TRAIN_BUCKET_MB", "32"))
        # with an RCCL communicator (comm) the collectives go through od_allreduce; without one but world_size > 1 they go
        # through torch.distributed's default group (gloo rehearsals on one GPU, CPU-side tests) -- same buckets, same streams
        need_c = (self.comm is not None or self.world > 1) and self.bucket_mb > 0
        self.cstream = (self._side_streams[1] or torch.cuda.Stream(device=dev)) if need_c else None
        self.payload_buf = (torch.empty(self.n_flat, dtype=torch.bfloat16, device=dev) if self.grad_payload == "bf16"
                            else None)
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)
        self._flag_queue, self._flag_pool = [], []  # (event, pinned host flag) of steps whose flag copy is still in flight
        self._consecutive_skips = 0
        self.max_consecutive_skips = 50  # fit() gives up after this many skipped steps in a row (see diverged())
        self._buckets = self._make_buckets(int(self.bucket_mb * (1 << 20) / 4)) if self.cstream is not None else []
        self._next_bucket = 0
        mxc = max(self.lib.od_bn_workspace_bytes(self.B * (n.H // n.stride) * (n.W // n.stride), n.Cout) + 2 * n.Cout * 4
                  for n in self.nodes)
        self.bn_ws = torch.empty(mxc, dtype=torch.uint8, device=dev)
        # BatchNorm statistics from the conv epilogue (od_conv_desc.bn_partials): one partial row per m-tile, at most M / 64
        # rows of 2 x Cout floats.  OFF by default: measured neutral (profiles/r02/train_bn_stats_fusion.txt: the separate
        # pass reads z hot out of L2 / MALL for 0.46 ms per step; the epilogue sums + more partial rows + giving up the
        # 8-wave kernel for those layers cost 0.5 ms).  OD_TRAIN_FUSE_BN_STATS=1 turns it on.
        self.fuse_bn_stats = os.environ.get("OD_TRAIN_FUSE_BN_STATS", "0") == "1"
        self._bn_rows = {}
        mxp = max(((self.B * (n.H // n.stride) * (n.W // n.stride) + 63) // 64) * 2 * n.Cout for n in self.nodes)
        self.bn_part = torch.empty(mxp, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------------------------------------------------
    def _make_buckets(self, min_elems):
        layers = []  # (lo, hi, name) per layer, in buffer order
        for name in self.specs:
            offs = [(o, o + (n + 3) // 4 * 4) for (nm, _k), (o, n) in self.seg.items() if nm == name]
            layers.append((min(o for o, _ in offs), max(h for _, h in offs), name))
        return make_buckets(layers, self.n_flat, min_elems)

    def _launch_ready_buckets(self, done_layers, main):
        """All-reduce every not-yet-launched bucket whose layers are all final (in order), on the communication stream."""
        while self._next_bucket < len(self._buckets):
            lo, hi, names = self._buckets[self._next_bucket]
            if not names <= done_layers:
                return
            self.cstream.wait_stream(main)  # BatchNorm / bias gradients of these layers
            if self.wstream is not None:
                self.cstream.wait_stream(self.wstream)  # their weight gradients (slab reduce)
            with torch.cuda.stream(self.cstream):
                self._reduce_range(lo, hi)
            self._next_bucket += 1

    def _reduce_range(self, lo, hi):
        """Sum grads[lo:hi] over the data-parallel ranks on the CURRENT stream, in the configured payload type."""
        lib, h, s = self.lib, self.ctx.handle, _stream_ptr()
        g = self.grads[lo:hi]
        if self.grad_payload == "bf16":
            pb = self.payload_buf[lo:hi]
            _lib.check(lib.od_cast_f32_bf16(h, g.data_ptr(), pb.data_ptr(), hi - lo, s), "od_cast_f32_bf16")
            if self.comm is not None:
                _lib.check(lib.od_allreduce(self.comm, pb.data_ptr(), hi - lo, _lib.OD_DT_BF16, s), "od_allreduce(bf16)")
            else:
                dp_allreduce_(pb)
            _lib.check(lib.od_cast_bf16_f32(h, pb.data_ptr(), g.data_ptr(), hi - lo, s), "od_cast_bf16_f32")
        elif self.comm is not None:
            _lib.check(lib.od_allreduce(self.comm, g.data_ptr(), hi - lo, _lib.OD_DT_F32, s), "od_allreduce")
        else:
            dp_allreduce_(g)

    def view(self, buf, name, kind):
        o, n = self.seg[(name, kind)]
        return buf[o:o + n]

    def _device_table(self, structs):
        """ctypes structs -> one device buffer (uploaded once; the multi-tensor kernels read it)"""
        raw = b"".join(bytes(st) for st in structs)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)

    def _repack(self):
        if getattr(self, "_pack_table", None) is None:
            layers = []
            for name, (_n, cin, cout, k, _s, _bn) in self.specs.items():
                if name == "b.conv0":
                    continue
                L = _lib.PackLayer()
                L.w_offset = self.seg[(name, "w")][0]
                L.w_fwd, L.w_bwd = self.wf[name].data_ptr(), self.wb[name].data_ptr()
                L.Cout, L.Cin, L.ksize = cout, cin, k
                layers.append(L)
            self._pack_table, self._pack_n = self._device_table(layers), len(layers)
        w0 = self.view(self.params, "b.conv0", "w")
        wf0 = self.wf["b.conv0"]
        wf0.zero_()
        wf0[:, :27] = w0.view(32, 27).to(torch.float16)  # 1.7 KB: plain torch copy is fine here
        _lib.check(self.lib.od_pack_weights_multi(self.ctx.handle, self.params.data_ptr(), self._pack_table.data_ptr(),
                                                  self._pack_n, _stream_ptr()), "od_pack_weights_multi")

    def _repack_per_layer(self):
        for name, (_n, cin, cout, k, _s, _bn) in self.specs.items():
            w = self.view(self.params, name, "w")
            if name == "b.conv0":
                wf = self.wf[name]
                wf.zero_()
                wf[:, :27] = w.view(32, 27).to(torch.float16)  # 1.7 KB: plain torch copy is fine here
                continue
            _lib.check(self.lib.od_pack_weights(self.ctx.handle, w.data_ptr(), self.wf[name].data_ptr(),
                                                self.wb[name].data_ptr(), cout, cin, k, _stream_ptr()), "od_pack_weights")

    def _add_node(self, name, x, out, H, Wd, act, res=None, res_mode="none", first=False, pred_off=None, pred_rows=0,
                  need_dx=True):
        _n, cin, cout, k, stride, bn = self.specs[name]
        n = _Node()
        n.name, n.x, n.out, n.res, n.res_mode = name, x, out, res, res_mode
        n.H, n.W, n.Cin, n.Cout, n.k, n.stride, n.act, n.bn = H, Wd, cin, cout, k, stride, act, bn
        n.first, n.pred_off, n.pred_rows, n.need_dx = first, pred_off, pred_rows, need_dx
        Ho, Wo = H // stride, Wd // stride
        dev = self.device
        if pred_off is None:
            n.z = torch.empty((self.B, Ho, Wo, cout), dtype=torch.float16, device=dev)
            self.tensors[out] = torch.empty((self.B, Ho, Wo, cout), dtype=torch.float16, device=dev)
            n.mean, n.rstd, n.scale, n.shift = (torch.empty(cout, dtype=torch.float32, device=dev) for _ in range(4))
        else:
            n.z = None
        self.nodes.append(n)
        return Ho, Wo

    def _build(self, bact, hact):
        H, Wd = self._add_node("b.conv0", "img", "t0", self.H0, self.W0, bact, first=True, need_dx=False)
        cur = "t0"
        taps = []
        for si, (nblk, _ch) in enumerate(W.STAGES, start=1):
            H, Wd = self._add_node(f"b.down{si}", cur, f"s{si}.d", H, Wd, bact)
            cur = f"s{si}.d"
            for r in range(nblk):
                self._add_node(f"b.s{si}.{r}.a", cur, f"s{si}.{r}.a", H, Wd, bact)
                self._add_node(f"b.s{si}.{r}.b", f"s{si}.{r}.a", f"s{si}.{r}.b", H, Wd, bact, res=cur, res_mode="same")
                cur = f"s{si}.{r}.b"
            taps.append((cur, H, Wd))
        (c3, h3, w3), (c4, h4, w4), (c5, h5, w5) = taps[2], taps[3], taps[4]
        self._add_node("n.lat5", c5, "p5", h5, w5, hact)
        self._add_node("n.lat4", c4, "m4", h4, w4, hact, res="p5", res_mode="up2")
        self._add_node("n.out4", "m4", "p4", h4, w4, hact)
        self._add_node("n.lat3", c3, "m3", h3, w3, hact, res="p4", res_mode="up2")
        self._add_node("n.out3", "m3", "p3", h3, w3, hact)
        off = 0
        for li, (lv, h, w) in enumerate((("p3", h3, w3), ("p4", h4, w4), ("p5", h5, w5))):
            cur = lv
            for t in range(self.tower):
                self._add_node(f"h.t{t}", cur, f"L{li}.t{t}", h, w, hact)
                cur = f"L{li}.t{t}"
            rows = h * w * W.NUM_PRIORS
            self._add_node("h.out", cur, f"L{li}.out", h, w, None, pred_off=off, pred_rows=rows)
            off += rows
        assert off == self.P

    # ------------------------------------------------------------------------------------------------------------
    def _conv_raw(self, n, x, out, out_f32=False, obs=0, ops=0, bias=None, bn_partials=None):
        d = _lib.ConvDesc()
        d.x, d.w = x.data_ptr(), self.wf[n.name].data_ptr()
        d.scale, d.bias = self.ones.data_ptr(), (bias if bias is not None else self.zeros).data_ptr()
        d.out = out if isinstance(out, int) else out.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = self.B, n.H, n.W, n.Cin, n.Cout, n.k, n.stride
        d.act, d.res_mode, d.tile_cfg = _lib.OD_ACT_LINEAR, _lib.OD_RES_NONE, self.fwd_tile_cfg
        d.out_dtype = _lib.OD_DT_F32 if out_f32 else _lib.OD_DT_F16
        d.out_batch_stride, d.out_pix_stride = obs, ops
        if bn_partials is not None:  # BatchNorm partial sums from the conv epilogue (no separate pass over z)
            d.bn_partials, d.bn_partials_bytes = bn_partials.data_ptr(), bn_partials.numel() * 4
            rows = self._bn_rows.get(n)
            if rows is None:
                rows = self._bn_rows[n] = self.lib.od_conv2d_fwd_bn_rows(self.ctx.handle, C.byref(d))
                if rows <= 0 or rows * 2 * n.Cout > bn_partials.numel():
                    raise _lib.OdError(f"bn_partials too small for {n.name}: {rows} rows")
        _lib.check(self.lib.od_conv2d_fwd(self.ctx.handle, C.byref(d), _stream_ptr()), f"conv fwd {n.name}")
        return self._bn_rows.get(n) if bn_partials is not None else None

    def forward(self, x_u8):
        """uint8 [B,H,W,3] (device) -> pred f32 [B,P,C]; keeps z / statistics of every layer for the backward pass."""
        self.tensors["img"].copy_(x_u8, non_blocking=True)
        lib, h = self.lib, self.ctx.handle
        s = _stream_ptr()
        self.run_stats_saved.copy_(self.run_stats, non_blocking=True)  # put back by sgd() when the step is skipped
        for n in self.nodes:
            x = self.tensors[n.x]
            if n.first:
                _lib.check(lib.od_conv_first_fwd(h, x.data_ptr(), self.wf[n.name].data_ptr(), self.in_scale.data_ptr(),
                                                 self.zeros.data_ptr(), n.z.data_ptr(), self.B, n.H, n.W, n.Cout,
                                                 _lib.OD_ACT_LINEAR, 0.0, s), "conv_first")
            elif n.pred_off is not None:
                bias = self.view(self.params, n.name, "bias")
                if bias.numel() < 256:  # scale/bias vectors are read in whole 256-channel pads
                    bp = getattr(self, "_bias_pad", None)
                    if bp is None:
                        bp = self._bias_pad = torch.zeros(256, dtype=torch.float32, device=self.device)
                    bp[:bias.numel()].copy_(bias)
                    bias = bp
                self._conv_raw(n, x, self.pred.data_ptr() + n.pred_off * self.C * 4, out_f32=True, obs=self.P * self.C,
                               ops=n.Cout, bias=bias)
                continue
            else:
                rows = self._conv_raw(n, x, n.z, bn_partials=self.bn_part if self.fuse_bn_stats else None)
            M = n.z.numel() // n.Cout
            if not n.first and self.fuse_bn_stats:
                _lib.check(lib.od_bn_stats_from_partials(h, self.bn_part.data_ptr(), rows, M, n.Cout,
                                                         self.view(self.params, n.name, "gamma").data_ptr(),
                                                         self.view(self.params, n.name, "beta").data_ptr(), W.BN_EPS,
                                                         n.mean.data_ptr(), n.rstd.data_ptr(), n.scale.data_ptr(),
                                                         n.shift.data_ptr(), self.run_mean[n.name].data_ptr(),
                                                         self.run_var[n.name].data_ptr(), BN_MOMENTUM, s),
                           "od_bn_stats_from_partials")
            else:
                wsb = lib.od_bn_workspace_bytes(M, n.Cout)
                _lib.check(lib.od_bn_stats(h, n.z.data_ptr(), M, n.Cout, self.view(self.params, n.name, "gamma").data_ptr(),
                                           self.view(self.params, n.name, "beta").data_ptr(), W.BN_EPS, n.mean.data_ptr(),
                                           n.rstd.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                                           self.run_mean[n.name].data_ptr(), self.run_var[n.name].data_ptr(), BN_MOMENTUM,
                                           self.bn_ws.data_ptr(), wsb, s), "od_bn_stats")
            y = self.tensors[n.out]
            res = self.tensors[n.res] if n.res else None
            rm = {"none": 0, "same": 1, "up2": 2}[n.res_mode]
            Ho, Wo = n.H // n.stride, n.W // n.stride
            _lib.check(lib.od_scale_act(h, n.z.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                                        res.data_ptr() if res is not None else None, rm, y.data_ptr(), self.B, Ho, Wo,
                                        n.Cout, _lib.ACT_ENUM[n.act[0] if n.act else None],
                                        float(n.act[1]) if n.act else 0.0, s), "od_scale_act")
        return self.pred

    def loss(self, y_target):
        """pred (from forward) + targets f32 [B,P,C] -> losses [4] on device; fills grad_pred."""
        _lib.check(self.lib.od_loss_fwd_bwd(self.ctx.handle, self.pred.data_ptr(), y_target.data_ptr(),
                                            self.grad_pred.data_ptr(), self.losses.data_ptr(), self.B, self.P,
                                            self.num_classes, 0.25, 2.0, {"smooth_l1": 0, "mse": 1}[self.box_mode], 1.0,
                                            1.0, 1.0, self.loss_ws.data_ptr(), self.loss_ws.numel(), _stream_ptr()),
                   "od_loss_fwd_bwd")
        return self.losses

    def _grad(self, key):
        t = self.gradbuf.get(key)
        if t is None:
            t = self.gradbuf[key] = torch.empty_like(self.tensors[key])
        return t

    def _build_wgrad_slabs(self):
        """Slab regions of the deterministic weight-gradient path: per layer, the slabs of all its nodes (a shared layer has
        one node per pyramid level) are contiguous so that one table entry sums them."""
        lib, h = self.lib, self.ctx.handle
        per_layer = {}
        for n in self.nodes:
            if n.first:
                continue
            sp = lib.od_conv2d_bwd_weight_splits(h, self.B, n.H, n.W, n.Cin, n.Cout, n.k, n.stride)
            assert sp > 0
            per_layer.setdefault(n.name, []).append((n, sp))
        total = sum(sum(sp for _n, sp in lst) * lst[0][0].Cout * lst[0][0].k ** 2 * lst[0][0].Cin for lst in per_layer.values())
        self.wgrad_slabs = torch.empty(total, dtype=torch.float32, device=self.device)
        self._slab_ptr, entries, off = {}, [], 0
        for name, lst in per_layer.items():
            n0 = lst[0][0]
            count = n0.Cout * n0.k ** 2 * n0.Cin
            e = _lib.WgradRed()
            e.dw_offset, e.count = self.seg[(name, "w")][0], count
            e.slabs = self.wgrad_slabs.data_ptr() + 4 * off
            e.nslabs = sum(sp for _n, sp in lst)
            entries.append(e)
            for n, sp in lst:
                self._slab_ptr[id(n)] = self.wgrad_slabs.data_ptr() + 4 * off
                off += sp * count
        self._wgrad_table, self._wgrad_n = self._device_table(entries), len(entries)
        # per-layer reduce right behind the layer's last weight-gradient launch (slabs still in L2 / MALL)
        self._wgrad_entry = {name: i for i, name in enumerate(per_layer)}
        self._wgrad_nodes = {name: len(lst) for name, lst in per_layer.items()}

    def backward(self):
        """grad_pred -> self.grads (f32, loss-scaled sums over this rank's batch)."""
        lib, h = self.lib, self.ctx.handle
        s = _stream_ptr()
        if getattr(self, "_wgrad_table", None) is None:
            self._build_wgrad_slabs()
        self.grads.zero_()
        have = set()
        pending = dict(self._wgrad_nodes)
        esz = C.sizeof(_lib.WgradRed)
        main = torch.cuda.current_stream(self.device)
        done_layers = set()
        self._next_bucket = 0
        if self.cstream is not None:
            self.cstream.wait_stream(main)  # grads.zero_() above
        for k, n in enumerate(reversed(self.nodes)):
            Ho, Wo = n.H // n.stride, n.W // n.stride
            M = self.B * Ho * Wo
            slot = k % len(self.dzs)
            if self._wg_done[slot] is not None:
                main.wait_event(self._wg_done[slot])  # the weight gradient that read this buffer three nodes ago
            dz = self.dzs[slot][:M * n.Cout]
            wsb = lib.od_bn_workspace_bytes(M, n.Cout) + 2 * n.Cout * 4
            if n.pred_off is not None:
                _lib.check(lib.od_pred_grad_to_level(h, self.grad_pred.data_ptr(), dz.data_ptr(), self.B, self.P, self.C,
                                                     n.pred_off, n.pred_rows, self.loss_scale, s), "od_pred_grad_to_level")
                _lib.check(lib.od_bn_bwd(h, dz.data_ptr(), dz.data_ptr(), self.ones.data_ptr(), self.zeros.data_ptr(),
                                         None, None, M, n.Cout, _lib.OD_ACT_LINEAR, 0.0, 0,
                                         self.scratch.data_ptr(), self.view(self.grads, n.name, "bias").data_ptr(),
                                         dz.data_ptr(), self.bn_ws.data_ptr(), wsb, s), "od_bn_bwd(bias)")
            else:
                assert n.out in have, f"no gradient reached {n.out}"
                dy = self.gradbuf[n.out]
                if n.res:
                    if n.res_mode == "same" and n.res not in have:
                        # first gradient to reach the shortcut's source: d(res) = dy.  No copy -- the source's gradient
                        # tensor IS dy's buffer from here on (dy has no reader after this node's od_bn_bwd below, every
                        # later contribution accumulates in place, all on this stream): a whole stage's residual stream
                        # shares one gradient buffer, and 22 full-tensor copies per step (0.4 ms at 32 x 320^2) are gone
                        self.gradbuf[n.res] = dy
                        g = dy
                    else:
                        g = self._grad(n.res)
                    if n.res_mode == "same":
                        if n.res in have:
                            _lib.check(lib.od_add_f16(h, g.data_ptr(), dy.data_ptr(), dy.numel(), s), "od_add_f16")
                    else:
                        _lib.check(lib.od_down2_sum_add(h, dy.data_ptr(), g.data_ptr(), self.B, Ho // 2, Wo // 2, n.Cout,
                                                        int(n.res in have), s), "od_down2_sum_add")
                    have.add(n.res)
                _lib.check(lib.od_bn_bwd(h, n.z.data_ptr(), dy.data_ptr(), n.scale.data_ptr(), n.shift.data_ptr(),
                                         n.mean.data_ptr(), n.rstd.data_ptr(), M, n.Cout,
                                         _lib.ACT_ENUM[n.act[0] if n.act else None], float(n.act[1]) if n.act else 0.0, 1,
                                         self.view(self.grads, n.name, "gamma").data_ptr(),
                                         self.view(self.grads, n.name, "beta").data_ptr(), dz.data_ptr(),
                                         self.bn_ws.data_ptr(), wsb, s), f"od_bn_bwd {n.name}")
            dw = self.view(self.grads, n.name, "w")
            x = self.tensors[n.x]
            # weight gradient: needs only dz and the saved input -> second stream, beside the dz -> dx -> ... chain
            ws = s
            if self.wstream is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                self.wstream.wait_event(ev)
                ws = C.c_void_p(self.wstream.cuda_stream)
            if n.first:
                if getattr(self, "_first_ws", None) is None:
                    nb = lib.od_conv_first_bwd_weight_workspace_bytes(h, self.B, n.H, n.W)
                    self._first_ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
                _lib.check(lib.od_conv_first_bwd_weight(h, x.data_ptr(), dz.data_ptr(), dw.data_ptr(), self.B, n.H, n.W,
                                                        n.Cout, 1.0 / 255.0, self._first_ws.data_ptr(),
                                                        self._first_ws.numel(), ws), "od_conv_first_bwd_weight")
            else:
                _lib.check(lib.od_conv2d_bwd_weight_slabs(h, x.data_ptr(), dz.data_ptr(), self._slab_ptr[id(n)], self.B, n.H,
                                                          n.W, n.Cin, n.Cout, n.k, n.stride, ws), f"wgrad {n.name}")
                pending[n.name] -= 1
                if pending[n.name] == 0:  # every node of the (possibly shared) layer has written its slabs: fixed-order sum
                    _lib.check(lib.od_wgrad_reduce_multi(h, self._wgrad_table.data_ptr() + esz * self._wgrad_entry[n.name],
                                                         1, self.grads.data_ptr(), ws), f"wgrad reduce {n.name}")
            if self.wstream is not None:
                done = torch.cuda.Event()
                done.record(self.wstream)
                self._wg_done[slot] = done
            if n.first or pending[n.name] == 0:
                done_layers.add(n.name)
                if self.cstream is not None:
                    self._launch_ready_buckets(done_layers, main)
            if n.first or not n.need_dx:
                continue
            g = self._grad(n.x)
            d = _lib.ConvDesc()
            d.x, d.w, d.scale, d.bias = dz.data_ptr(), self.wb[n.name].data_ptr(), self.ones.data_ptr(), self.zeros.data_ptr()
            d.out = g.data_ptr()
            d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = self.B, Ho, Wo, n.Cout, n.Cin, n.k, n.stride
            d.act, d.out_dtype, d.tile_cfg = _lib.OD_ACT_LINEAR, _lib.OD_DT_F16, self.bwd_tile_cfg
            d.transposed = int(n.stride == 2)
            if n.x in have:
                d.res, d.res_mode = g.data_ptr(), _lib.OD_RES_SAME  # accumulate in place
            else:
                d.res, d.res_mode = None, _lib.OD_RES_NONE
            _lib.check(lib.od_conv2d_fwd(h, C.byref(d), s), f"dgrad {n.name}")
            have.add(n.x)
        if self.wstream is not None:
            main.wait_stream(self.wstream)
            self._wg_done = [None] * len(self.dzs)
        # (conv weight gradients: each layer's per-split slabs were summed in a fixed order right behind its last
        # weight-gradient launch -- no atomics, and the slabs are still cache-resident when they are read back)
        # shared-layer BN gradients were accumulated over the three levels inside od_bn_bwd
        return self.grads

    def allreduce(self):
        """Sum the flat f32 gradient buffer over the data-parallel ranks: bucketed all-reduces launched during backward on
        the communication stream (RCCL through the C ABI), or one collective after backward (47 M floats)."""
        if self.cstream is not None:  # bucketed: every bucket was launched during backward, join the communication stream
            assert self._next_bucket == len(self._buckets), "a gradient bucket was never launched"
            torch.cuda.current_stream(self.device).wait_stream(self.cstream)
            return
        if self.world <= 1 and self.comm is None:
            return
        self._reduce_range(0, self.n_flat)

    def sgd(self):
        """One multi-tensor launch over the flat parameter buffer (per-segment LR multipliers of docs/MODEL.md:84-90 and
        weight decay in a device table), one multi-layer re-pack launch."""
        inv = dp_effective_scale(self.loss_scale, self.world)  # grads are averaged over ranks
        # after the all-reduce, so every rank sees the same flag: Inf / NaN anywhere -> this step's update is skipped
        _lib.check(self.lib.od_grad_nonfinite(self.ctx.handle, self.grads.data_ptr(), self.n_flat, self.nonfinite.data_ptr(),
                                              _stream_ptr()), "od_grad_nonfinite")
        key = (self.lr, self.weight_decay, tuple(sorted(self.lr_multipliers.items())))
        if getattr(self, "_sgd_key", None) != key:
            segs = []
            for (name, kind), (o, n) in self.seg.items():
                sg = _lib.SgdSeg()
                sg.offset, sg.count = o, n
                sg.lr = self.lr * lr_multiplier(name, self.lr_multipliers)
                sg.weight_decay = self.weight_decay if kind == "w" else 0.0
                segs.append(sg)
            self._sgd_table, self._sgd_n, self._sgd_key = self._device_table(segs), len(segs), key
        _lib.check(self.lib.od_sgd_step_multi(self.ctx.handle, self.params.data_ptr(), self.mom.data_ptr(),
                                              self.grads.data_ptr(), self._sgd_table.data_ptr(), self._sgd_n,
                                              self.momentum, inv, self.nonfinite.data_ptr(), _stream_ptr()),
                   "od_sgd_step_multi")
        # a skipped step also takes back what its forward pass did to the BatchNorm running statistics: when the Inf / NaN
        # came from an f16 overflow in z (not from a loss-scaled dz) the batch statistics folded into them are non-finite
        _lib.check(self.lib.od_copy_if_nonzero(self.ctx.handle, self.run_stats.data_ptr(), self.run_stats_saved.data_ptr(),
                                               self.run_stats.numel(), self.nonfinite.data_ptr(), _stream_ptr()),
                   "od_copy_if_nonzero")
        self._repack()  # (a skipped step re-packs unchanged masters: harmless, and keeps the launch sequence fixed)
        host = self._flag_pool.pop() if self._flag_pool else torch.zeros(1, dtype=torch.int32).pin_memory()
        host.copy_(self.nonfinite, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._flag_queue.append((ev, host))

    def _poll_nonfinite(self, wait=False):
        """Consume the flags of earlier steps whose device->host copy HAS completed (event.query(): the host never waits
        for the previous step here, so encode_batch and the first launches of this step overlap its tail; the loss scale
        reacts one or two steps late).  A skipped step halves the loss scale; a long clean run doubles it again.
        wait=True drains the queue (end of fit(), tests)."""
        while self._flag_queue:
            ev, host = self._flag_queue[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                break
            self._flag_queue.pop(0)
            bad = int(host[0]) != 0
            self._flag_pool.append(host)
            self._consecutive_skips = self._consecutive_skips + 1 if bad else 0
            if bad:
                self.skipped_steps += 1
                self._good_steps = 0
                if self.dynamic_loss_scale:
                    self.loss_scale = max(1.0, self.loss_scale * 0.5)
            else:
                self._good_steps += 1
                if self.dynamic_loss_scale and self._good_steps >= self.loss_scale_growth_interval:
                    self.loss_scale, self._good_steps = min(self.loss_scale * 2.0, 65536.0), 0

    def diverged(self):
        """True when the last `max_consecutive_skips` steps were ALL skipped although the loss scale has already been halved
        as far as it can help: the FORWARD pass overflows f16 (the weights themselves are in a bad place -- typically a
        learning rate too high for the batch size), which no loss scale can cure and a skipped step never changes."""
        return self._consecutive_skips >= self.max_consecutive_skips

    def sgd_per_tensor(self):
        """Per-tensor form of sgd() (kept for the equality test of the multi-tensor kernels).  UNGUARDED: no non-finite
        check, no loss-scale update, no restore of the running statistics -- training loops use sgd()."""
        inv = dp_effective_scale(self.loss_scale, self.world)  # grads are averaged over ranks
        for (name, kind), (o, n) in self.seg.items():
            lr = self.lr * lr_multiplier(name, self.lr_multipliers)
            wd = self.weight_decay if kind == "w" else 0.0
            _lib.check(self.lib.od_sgd_step(self.ctx.handle, self.params.data_ptr() + 4 * o, self.mom.data_ptr() + 4 * o,
                                            self.grads.data_ptr() + 4 * o, n, lr, self.momentum, wd, inv, _stream_ptr()),
                       "od_sgd_step")
        self._repack_per_layer()

    def step(self, x_u8, annotations=None, y_target=None):
        """One training step.  annotations: list[ObjectsAnnotation] (encoded on the device) or y_target [B,P,C]."""
        self._poll_nonfinite()
        if y_target is None:
            y_target, _npos, _ = self.pb.encode_batch(annotations, return_device=True)
        self.forward(x_u8)
        self.loss(y_target)
        self.backward()
        self.allreduce()
        self.sgd()
        return self.losses

    def fit(self, batches, steps, lr_schedule=None, log_every=0, log=print):
        """Run `steps` training steps from an iterator of (x_u8 [B,H,W,3] device tensor, y_target [B,P,C] device tensor or
        list of annotations) -- what od_gen.Generator(on_device=True).flow(...) yields (the reference's unseen `fit`:
        generator -> encode_truth -> losses, check_assign.py:21-22).  lr_schedule(step) -> learning rate.  Returns the
        per-step losses [steps, 4] (objectness, class, box, total) as a numpy array; losses stay on the device until the
        end, so the loop never synchronises."""
        hist = torch.zeros((steps, 4), dtype=torch.float32, device=self.device)
        for i, (xb, yb) in zip(range(steps), batches):
            if lr_schedule is not None:
                self.lr = float(lr_schedule(i))
            if isinstance(yb, torch.Tensor):
                self.step(xb, y_target=yb)
            else:
                self.step(xb, annotations=yb)
            hist[i].copy_(self.losses)
            if self.diverged():
                raise RuntimeError(
                    f"training diverged: {self._consecutive_skips} consecutive steps produced non-finite values (step {i + 1}, loss "
                    f"scale {self.loss_scale:g}); the forward pass itself overflows f16 -- lower the learning rate (it scales "
                    f"with the batch size) or lengthen the warm-up; the weights of the last good step are intact")
            if log_every and (i + 1) % log_every == 0:
                l = hist[i].cpu().numpy()
                log(f"step {i + 1}/{steps} lr {self.lr:.4g} loss {l[3]:.4f} (obj {l[0]:.4f} cls {l[1]:.4f} box {l[2]:.4f}) "
                    f"loss_scale {self.loss_scale:g} skipped {self.skipped_steps}")
        self._poll_nonfinite(wait=True)
        return hist.cpu().numpy()

    def export_params(self):
        """-> dict in the weights.py format (masters + BatchNorm running statistics) for ObjectDetector(params, ...)."""
        host = self.params.cpu().numpy()
        out = {}
        for (name, kind), (o, n) in self.seg.items():
            _n, cin, cout, k, _s, _bn = self.specs[name]
            a = host[o:o + n]
            out[f"{name}.{kind}"] = a.reshape(cout, k, k, cin).copy() if kind == "w" else a.copy()
        for name in self.run_mean:
            out[name + ".mean"] = self.run_mean[name].cpu().numpy()
            out[name + ".var"] = self.run_var[name].cpu().numpy()
        return out


def make_buckets(layers, n_flat, min_elems):
    """Gradient buckets for the overlapped all-reduce.  layers: [(lo, hi, name)] in buffer (= forward) order, contiguous
    and covering [0, n_flat).  -> [(lo, hi, {layer names})] in the order backward completes them: contiguous ranges cut
    at layer boundaries, walking from the last layer to the first, each at least `min_elems` long (the last one takes
    whatever is left)."""
    buckets, hi, names = [], n_flat, set()
    for lo, _h, name in reversed(layers):
        names.add(name)
        if hi - lo >= min_elems:
            buckets.append((lo, hi, names))
            hi, names = lo, set()
    if names:
        buckets.append((0, hi, names))
    return buckets


def dp_allreduce_(flat: torch.Tensor):
    """In-place sum over the ranks of the default process group (backend "nccl" = RCCL on GPUs, gloo in CPU tests)."""
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    return flat


def dp_effective_scale(loss_scale: float, world: int) -> float:
    """od_sgd_step's inv_loss_scale: undo the loss scale and average the summed gradients over the ranks."""
    return 1.0 / (loss_scale * world)


def init_comm(ctx: Context):
    """Create the RCCL communicator of this rank through the C ABI; the unique id travels over torch.distributed."""
    rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    nbytes = ctx.lib.od_comm_unique_id_bytes()
    buf = (C.c_ubyte * nbytes)()
    if rank == 0:
        _lib.check(ctx.lib.od_comm_get_unique_id(buf, nbytes), "od_comm_get_unique_id")
    box = [bytes(buf)]
    torch.distributed.broadcast_object_list(box, src=0)
    raw = (C.c_ubyte * nbytes).from_buffer_copy(box[0])
    h = C.c_void_p()
    _lib.check(ctx.lib.od_comm_init(ctx.handle, rank, world, raw, C.byref(h)), "od_comm_init")
    return h, world
