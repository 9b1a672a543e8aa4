"""Device-side network: packs parameters for the HIP kernels, owns the activation buffers (torch tensors are only
containers of HBM) and builds the native forward plan executed by libodhip.so.

Replaces the Keras model graph that reference `ObjectDetector.load_voc(...)` builds and `predict` runs
(voc_validate.py:25-27): Darknet53 conv/BN/LeakyReLU stack, FPN-like neck, shared prediction module
(docs/MODEL.md:5-27).  HBM layout: activations NHWC f16, one buffer per layer output; the prediction convs of the
three levels write f32 straight into their slice of pred[B, P, 2+NC+4] (level-major, then y, x, prior).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib, weights as W


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Context:
    """od_ctx wrapper (one per process / device)."""

    _by_device = {}

    def __init__(self, device_index: int):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.od_ctx_create(device_index, C.byref(h)), "od_ctx_create")
        self.handle = h
        self.device_index = device_index

    @classmethod
    def get(cls, device) -> "Context":
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.OdError(f"object_detector_amd runs on MI355X only (got device {dev}); there is no CPU path")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if idx not in cls._by_device:
            cls._by_device[idx] = cls(idx)
        return cls._by_device[idx]


def pack_conv_weight(w_ohwi: np.ndarray):
    """[Cout,k,k,Cin] -> f16 [Cout_pad, Kpad], k index = (dy*k+dx)*Cin + cin (the implicit-GEMM K order)."""
    cout, k, _, cin = w_ohwi.shape
    cout_pad, kpad = _lib.conv_weight_dims(cout, cin, k)
    out = np.zeros((cout_pad, kpad), np.float16)
    out[:cout, :k * k * cin] = w_ohwi.reshape(cout, k * k * cin).astype(np.float16)
    return out


def pack_first_weight(w_ohwi: np.ndarray):
    cout = w_ohwi.shape[0]
    out = np.zeros((cout, 32), np.float16)
    out[:, :27] = w_ohwi.reshape(cout, 27).astype(np.float16)
    return out


def pad_vec(v: np.ndarray, n: int):
    out = np.zeros(n, np.float32)
    out[:len(v)] = v
    return out


class Net:
    def __init__(self, params, batch_size, input_size=(320, 320), device="cuda:0", backbone_act=("leaky", 0.1),
                 head_act=("elu", 1.0), tile_cfg=None, overlapped=False, share_weights_with=None):
        self.ctx = Context.get(device)
        self.lib = self.ctx.lib
        self.device = torch.device(device)
        self.B = int(batch_size)
        self.H, self.W = int(input_size[0]), int(input_size[1])
        if self.H % 32 or self.W % 32:
            raise ValueError("input_size must be a multiple of 32 (three maps at stride 8/16/32)")
        self.num_classes, self.neck_ch, self.tower = W.infer_arch(params)
        self.C = 2 + self.num_classes + 4
        self.level_hw = [(self.H // s, self.W // s) for s in (8, 16, 32)]
        self.P = sum(h * w for h, w in self.level_hw) * W.NUM_PRIORS
        self.tile_cfg = dict(tile_cfg or {})
        self.auto_cfg = -2 if overlapped else -1  # od_conv_desc.tile_cfg: -2 = this plan runs beside other batches in flight
        for item in filter(None, os.environ.get("OD_TILE_CFG", "").split(",")):  # tuning: "b.s3=24,n.lat=25" (name prefixes)
            pat, cfg = item.split("=")
            self.tile_cfg[pat] = int(cfg)
        self.fuse_blocks = os.environ.get("OD_FUSE_BLOCKS", "1") != "0"  # fused residual blocks of the early stages
        self.splitk = True  # small-M layers (batch-1) may use split-K through a shared f32 slab workspace
        self._splitk_elems = 0
        self._splitk_descs = []
        self._keep = []  # device tensors referenced by raw pointers in the plan
        self._dev = {}
        if share_weights_with is not None:  # another pipeline of the same detector: the packed weights are read-only
            self._dev = share_weights_with._dev
        for name, cin, cout, k, _s, _bn in ([] if share_weights_with is not None else
                                            W.layer_specs(self.num_classes, self.neck_ch, self.tower)):
            w = params[name + ".w"]
            assert w.shape == (cout, k, k, cin), (name, w.shape)
            scale, bias = W.fold_bn(params, name)
            if name == "b.conv0":
                wp = pack_first_weight(w)
                scale = (scale / np.float32(255.0)).astype(np.float32)  # uint8 input normalisation folded in
                npad = cout
            else:
                wp = pack_conv_weight(w)
                npad = wp.shape[0]
            self._dev[name] = (self._to_dev(wp), self._to_dev(pad_vec(scale, npad)), self._to_dev(pad_vec(bias, npad)))
        self.input = torch.zeros((self.B, self.H, self.W, 3), dtype=torch.uint8, device=self.device)
        self.pred = torch.zeros((self.B, self.P, self.C), dtype=torch.float32, device=self.device)
        self.ops = []       # (_lib.PlanOp)
        self.op_info = []   # dict(name, flops, bytes)
        self._build(backbone_act, head_act)
        if self._splitk_elems:
            nbytes = min(32 * self._splitk_elems * 4, 256 << 20)  # room for up to 32 partial slabs of the largest layer
            self.splitk_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            for i in self._splitk_descs:
                self.ops[i].conv.splitk = 0
                self.ops[i].conv.splitk_workspace = self.splitk_ws.data_ptr()
                self.ops[i].conv.splitk_workspace_bytes = nbytes
        # TIMING ABLATION ONLY (results become garbage): drop ops whose name matches a prefix*suffix pattern, to measure what a
        # group of layers costs inside the running pipeline (scripts/dev/exp_ablate.sh); never set in tests / bench lines
        abl = [q.partition("*") for q in filter(None, os.environ.get("OD_ABLATE_OPS", "").split(","))]
        if abl:
            seen, occ = {}, []
            for inf in self.op_info:  # "name#k" = the k-th launch of a layer that runs once per pyramid level
                occ.append(f'{inf["name"]}#{seen.get(inf["name"], 0)}')
                seen[inf["name"]] = seen.get(inf["name"], 0) + 1
            keep = [i for i, inf in enumerate(self.op_info)
                    if not any((inf["name"].startswith(pre) and inf["name"].endswith(suf)) or occ[i] == pre + _s + suf
                               for pre, _s, suf in abl)]
            self.ops = [self.ops[i] for i in keep]
            self.op_info = [self.op_info[i] for i in keep]
        arr = (_lib.PlanOp * len(self.ops))(*self.ops)
        h = C.c_void_p()
        _lib.check(self.lib.od_plan_create(self.ctx.handle, arr, len(self.ops), C.byref(h)), "od_plan_create")
        self.plan = h
        self._captured = False

    # ------------------------------------------------------------------------------------------------
    def _to_dev(self, a: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self._keep.append(t)
        return t

    def _buf(self, h, w, c):
        t = torch.empty((self.B, h, w, c), dtype=torch.float16, device=self.device)
        self._keep.append(t)
        return t

    def _conv(self, name, x, h, w, cin, cout, k, stride, act, res=None, res_mode=_lib.OD_RES_NONE, out=None,
              out_f32=False, obs=0, ops=0, then=None):
        """then = (name2, cout2, act2): the pointwise layer that consumes this layer's output rides in the same op
        (od_conv_desc.w2: inside the 8-wave kernel's epilogue, or as a second launch of the library's choosing);
        returns (out, ho, wo, out2) then."""
        wt, sc, bi = self._dev[name]
        ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
        if out is None:
            out = self._buf(ho, wo, cout)
        d = _lib.ConvDesc()
        d.x, d.w, d.scale, d.bias = x.data_ptr(), wt.data_ptr(), sc.data_ptr(), bi.data_ptr()
        d.res = res.data_ptr() if res is not None else None
        d.out = out if isinstance(out, int) else out.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cout = self.B, h, w, cin, cout
        d.ksize, d.stride = k, stride
        d.act, d.alpha = _lib.ACT_ENUM[act[0] if act else None], float(act[1]) if act else 0.0
        d.res_mode = res_mode
        d.out_dtype = _lib.OD_DT_F32 if out_f32 else _lib.OD_DT_F16
        d.out_batch_stride, d.out_pix_stride = obs, ops
        d.tile_cfg = self.tile_cfg.get(name, self.auto_cfg)
        for pat, cfg in self.tile_cfg.items():  # patterns "prefix*suffix", e.g. "b.s3*a" = the 1x1 convs of stage 3
            pre, star, suf = pat.partition("*")
            if star and name.startswith(pre) and name.endswith(suf):
                d.tile_cfg = cfg
        out2 = None
        if then is not None:
            name2, cout2, act2 = then
            w2, sc2, bi2 = self._dev[name2]
            out2 = self._buf(ho, wo, cout2)
            d.w2, d.scale2, d.bias2, d.out2 = w2.data_ptr(), sc2.data_ptr(), bi2.data_ptr(), out2.data_ptr()
            d.Cout2 = cout2
            d.act2, d.alpha2 = _lib.ACT_ENUM[act2[0] if act2 else None], float(act2[1]) if act2 else 0.0
        m = self.B * ho * wo
        if self.splitk and m * cout <= (1 << 22):  # candidates only; the library decides per layer
            self._splitk_elems = max(self._splitk_elems, m * cout)
            self._splitk_descs.append(len(self.ops))
        op = _lib.PlanOp()
        op.kind = _lib.OD_OP_CONV
        op.conv = d
        self.ops.append(op)
        osz = 4 if out_f32 else 2
        self.op_info.append(dict(name=name, flops=2.0 * m * cout * k * k * cin,
                                 bytes=float(self.B * h * w * cin * 2 + m * cout * osz + cout * k * k * cin * 2
                                             + (m * cout * 2 if res_mode == _lib.OD_RES_SAME else 0)
                                             + (m * cout // 2 if res_mode == _lib.OD_RES_UP2 else 0)),
                                 shape=(m, cout, k * k * cin),
                                 # the op's tensors, for per-layer parity checks (tests/test_gpu_fullsize.py)
                                 kind="conv", x=x, res=res, res_mode=res_mode, out=None if isinstance(out, int) else out,
                                 stride=stride, ksize=k, act=act, out_f32=out_f32))
        if then is not None:
            inf = self.op_info[-1]
            inf["flops"] += 2.0 * m * cout2 * cout
            inf["bytes"] += float(m * cout2 * 2 + cout2 * cout * 2)  # the 1x1's input never comes back from HBM
            inf["then"] = dict(name=name2, out=out2, act=act2)
            return out, ho, wo, out2
        return out, ho, wo

    def _bneck(self, name, x, h, w, ch, act):
        """one fused residual block (1x1 ch -> ch/2, 3x3 ch/2 -> ch, + x): od_bottleneck_fwd"""
        w1, s1, b1 = self._dev[name + ".a"]
        w3, s3, b3 = self._dev[name + ".b"]
        out = self._buf(h, w, ch)
        d = _lib.BneckDesc()
        d.x, d.out = x.data_ptr(), out.data_ptr()
        d.w1, d.scale1, d.bias1 = w1.data_ptr(), s1.data_ptr(), b1.data_ptr()
        d.w3, d.scale3, d.bias3 = w3.data_ptr(), s3.data_ptr(), b3.data_ptr()
        d.B, d.H, d.W, d.C = self.B, h, w, ch
        d.act, d.alpha = _lib.ACT_ENUM[act[0] if act else None], float(act[1]) if act else 0.0
        op = _lib.PlanOp()
        op.kind = _lib.OD_OP_BNECK
        op.bneck = d
        self.ops.append(op)
        m = self.B * h * w
        self.op_info.append(dict(name=name, flops=2.0 * m * (ch * (ch // 2) + 9 * (ch // 2) * ch),
                                 bytes=float(2 * m * ch * 2 + (ch * ch // 2 + 9 * ch * ch // 2) * 2),
                                 shape=(m, ch, 10 * (ch // 2)), kind="bneck", x=x, out=out, act=act))
        return out

    def _build(self, bact, hact):
        B, H, Wd = self.B, self.H, self.W
        fuse_stem = (self.fuse_blocks and os.environ.get("OD_FUSE_STEM", "1") != "0"
                     and self.lib.od_stem_supported(H, Wd))
        if fuse_stem:
            # first two layers in one launch (od_stem_fwd): uint8 in, f16 [B,H/2,W/2,64] out
            w0, sc0, bi0 = self._dev["b.conv0"]
            w3, sc3, bi3 = self._dev["b.down1"]
            x = self._buf(H // 2, Wd // 2, 64)
            d = _lib.StemDesc()
            d.x, d.out = self.input.data_ptr(), x.data_ptr()
            d.w0, d.scale0, d.bias0 = w0.data_ptr(), sc0.data_ptr(), bi0.data_ptr()
            d.w3, d.scale3, d.bias3 = w3.data_ptr(), sc3.data_ptr(), bi3.data_ptr()
            d.B, d.H, d.W = B, H, Wd
            d.act, d.alpha = _lib.ACT_ENUM[bact[0]], float(bact[1])
            op = _lib.PlanOp()
            op.kind = _lib.OD_OP_STEM
            op.stem = d
            self.ops.append(op)
            m1 = B * (H // 2) * (Wd // 2)
            self.op_info.append(dict(name="b.stem", flops=2.0 * B * H * Wd * 32 * 27 + 2.0 * m1 * 64 * 288,
                                     bytes=float(B * H * Wd * 3 + m1 * 64 * 2), shape=(m1, 64, 288 + 27),
                                     kind="stem", x=self.input, out=x, act=bact))
        else:
            # first layer (uint8 in)
            wt, sc, bi = self._dev["b.conv0"]
            x = self._buf(H, Wd, 32)
            d = _lib.ConvDesc()
            d.x, d.w, d.scale, d.bias, d.out = (self.input.data_ptr(), wt.data_ptr(), sc.data_ptr(), bi.data_ptr(),
                                                x.data_ptr())
            d.B, d.H, d.W, d.Cin, d.Cout = B, H, Wd, 3, 32
            d.ksize, d.stride = 3, 1
            d.act, d.alpha = _lib.ACT_ENUM[bact[0]], float(bact[1])
            op = _lib.PlanOp()
            op.kind = _lib.OD_OP_CONV_FIRST
            op.conv = d
            self.ops.append(op)
            self.op_info.append(dict(name="b.conv0", flops=2.0 * B * H * Wd * 32 * 27,
                                     bytes=float(B * H * Wd * (3 + 64)), shape=(B * H * Wd, 32, 27),
                                     kind="first", x=self.input, out=x, act=bact))
        h, w, cin = H, Wd, 32
        taps = []
        for si, (n, ch) in enumerate(W.STAGES, start=1):
            fused_block = (self.fuse_blocks and os.environ.get(f"OD_FUSE_BNECK{ch}", "1") != "0"
                           and self.lib.od_bottleneck_supported(h // 2, w // 2, ch))
            # 256-channel stage: a block's 1x1 (256 -> 128) rides in the launch that PRODUCES its input (the stride-2 conv
            # or the previous block's 3x3 -- all 256 channels of a pixel are in one workgroup of the 8-wave kernel)
            ride = self.fuse_blocks and ch == 256 and not fused_block and os.environ.get("OD_FUSE_POINTWISE", "1") != "0"
            t = None
            if si == 1 and fuse_stem:
                h, w = h // 2, w // 2
            elif ride:
                x, h, w, t = self._conv(f"b.down{si}", x, h, w, cin, ch, 3, 2, bact, then=(f"b.s{si}.0.a", ch // 2, bact))
            else:
                x, h, w = self._conv(f"b.down{si}", x, h, w, cin, ch, 3, 2, bact)
            for r in range(n):
                if fused_block:
                    x = self._bneck(f"b.s{si}.{r}", x, h, w, ch, bact)
                    continue
                if t is None:
                    t, _, _ = self._conv(f"b.s{si}.{r}.a", x, h, w, ch, ch // 2, 1, 1, bact)
                if ride and r + 1 < n:
                    x, _, _, t = self._conv(f"b.s{si}.{r}.b", t, h, w, ch // 2, ch, 3, 1, bact, res=x,
                                            res_mode=_lib.OD_RES_SAME, then=(f"b.s{si}.{r + 1}.a", ch // 2, bact))
                else:
                    x, _, _ = self._conv(f"b.s{si}.{r}.b", t, h, w, ch // 2, ch, 3, 1, bact, res=x,
                                         res_mode=_lib.OD_RES_SAME)
                    t = None
            cin = ch
            taps.append((x, h, w, ch))
        (c3, h3, w3, ch3), (c4, h4, w4, ch4), (c5, h5, w5, ch5) = taps[2], taps[3], taps[4]
        nc = self.neck_ch
        p5, _, _ = self._conv("n.lat5", c5, h5, w5, ch5, nc, 1, 1, hact)
        m4, _, _ = self._conv("n.lat4", c4, h4, w4, ch4, nc, 1, 1, hact, res=p5, res_mode=_lib.OD_RES_UP2)
        p4, _, _ = self._conv("n.out4", m4, h4, w4, nc, nc, 3, 1, hact)
        m3, _, _ = self._conv("n.lat3", c3, h3, w3, ch3, nc, 1, 1, hact, res=p4, res_mode=_lib.OD_RES_UP2)
        p3, _, _ = self._conv("n.out3", m3, h3, w3, nc, nc, 3, 1, hact)
        self.levels = [(p3, h3, w3), (p4, h4, w4), (p5, h5, w5)]
        self.taps = [c3, c4, c5]
        # shared prediction module; the last conv writes f32 logits into pred[:, off:off+h*w*8, :]
        off = 0
        cout = W.NUM_PRIORS * self.C
        for (x, h, w) in self.levels:
            t = x
            for i in range(self.tower):
                t, _, _ = self._conv(f"h.t{i}", t, h, w, nc, nc, 3, 1, hact)
            out_ptr = self.pred.data_ptr() + off * self.C * 4
            self._conv("h.out", t, h, w, nc, cout, 3, 1, None, out=out_ptr, out_f32=True,
                       obs=self.P * self.C, ops=cout)
            self.op_info[-1]["pred_rows"] = (off, h * w * W.NUM_PRIORS)
            off += h * w * W.NUM_PRIORS
        assert off == self.P

    # ------------------------------------------------------------------------------------------------
    def forward(self, x_u8: torch.Tensor | None = None, graph: bool = False) -> torch.Tensor:
        """uint8 [B,H,W,3] device tensor (or None = reuse self.input) -> pred f32 [B,P,2+NC+4] (owned by the Net)."""
        if x_u8 is not None:
            if x_u8.shape != self.input.shape or x_u8.dtype != torch.uint8:
                raise ValueError(f"expected uint8 {tuple(self.input.shape)}, got {x_u8.dtype} {tuple(x_u8.shape)}")
            self.input.copy_(x_u8, non_blocking=True)
        if graph:
            if not self._captured:
                s = torch.cuda.Stream(device=self.device)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    _lib.check(self.lib.od_plan_capture(self.plan, C.c_void_p(s.cuda_stream)), "od_plan_capture")
                torch.cuda.current_stream().wait_stream(s)
                self._captured = True
            _lib.check(self.lib.od_plan_replay(self.plan, _stream_ptr()), "od_plan_replay")
        else:
            _lib.check(self.lib.od_plan_run(self.plan, _stream_ptr()), "od_plan_run")
        return self.pred

    def time_ops(self):
        """Per-op device time (ms) via hipEvents on the launch stream + kernel names: for bench.py's roofline."""
        n = len(self.ops)
        ms = (C.c_float * n)()
        _lib.check(self.lib.od_plan_time_ops(self.plan, _stream_ptr(), ms, n), "od_plan_time_ops")
        names = [self.lib.od_plan_op_kernel_name(self.plan, i).decode() for i in range(n)]
        return list(ms), names

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.od_plan_destroy(self.plan)
        except Exception:
            pass
