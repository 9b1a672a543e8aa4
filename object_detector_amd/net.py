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


# the cheapest plan that keeps EVERY logit within 1e-3 x scale of the fp32 oracle at 32 x 320^2 / 16 x 640^2 with margin
# (profiles/r03/logit_error_mixed_plans_cpu.txt: 22 % of the default plan's error variance; the stage-3 stream would cost
# another 0.29 ms per batch for 6 points of variance that splitting the two small lateral layers buys for nothing)
MIXED_STREAM_STAGES = (4, 5)
MIXED_SPLIT = ("n.lat4", "n.lat5", "n.out3", "n.out4", "h.t0", "h.out")
# who reads a neck / head tensor as a convolution operand (producer layer -> consumer layers)
_NECK_CONSUMERS = {"n.lat5": ("h.t0",), "n.lat4": ("n.out4",), "n.out4": ("h.t0",), "n.lat3": ("n.out3",), "n.out3": ("h.t0",),
                   "h.t0": ("h.out",)}


class Net:
    """precision="f16" (default): every stored activation is f16 -- the throughput plan.
    precision="mixed": north_star's "within 1e-3 on logits" at ANY logit scale (measured attribution: DESIGN.md §5,
    scripts/dev/attribute_logit_error*.py): the residual stream of `stream_stages` is kept in f32 (od_wide_add: x32 += y32,
    the next conv's f16 operand is rounded once from the f32 sum) and the convolutions in `split` read their input as an f16
    (hi, lo) pair (2x the MFMA work of those layers, ~22 significant bits).  Every multiply is still an f16 MFMA."""

    def __init__(self, params, batch_size, input_size=(320, 320), device="cuda:0", backbone_act=("leaky", 0.1),
                 head_act=("elu", 1.0), tile_cfg=None, overlapped=False, share_weights_with=None, precision=None,
                 stream_stages=None, split=None, wide_fpn=None):
        self.wide_fpn = (os.environ.get("OD_MIXED_WIDE_FPN", "1") != "0") if wide_fpn is None else bool(wide_fpn)
        self.precision = precision or os.environ.get("OD_PRECISION", "f16")
        if self.precision not in ("f16", "mixed"):
            raise ValueError(f"precision must be 'f16' or 'mixed', got {self.precision!r}")
        env_st, env_sp = os.environ.get("OD_MIXED_STREAM"), os.environ.get("OD_MIXED_SPLIT")
        self.stream_stages = tuple(stream_stages if stream_stages is not None else
                                   ([int(v) for v in env_st.split(",") if v] if env_st is not None else MIXED_STREAM_STAGES))
        self.split = tuple(split if split is not None else
                           ([v for v in env_sp.split(",") if v] if env_sp is not None else MIXED_SPLIT))
        if self.precision == "f16":
            self.stream_stages, self.split, self.wide_fpn = (), (), False
        if any(k not in (3, 4, 5) for k in self.stream_stages):
            raise ValueError("stream_stages: stages 3, 4, 5 only (stages 1-2 run as fused residual blocks)")
        for nm in self.split:
            if nm not in ("n.lat3", "n.lat4", "n.lat5", "n.out3", "n.out4", "h.t0", "h.out"):
                raise ValueError(f"split: {nm!r} is not a neck / prediction-module layer")
            if nm.startswith("n.lat") and int(nm[-1]) not in self.stream_stages:
                raise ValueError(f"split layer {nm} reads the stage-{nm[-1]} tap: that stage must be in stream_stages")
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
            if name in self.split:  # operand = [hi | lo] along the channels: the same weights for both halves
                self._dev[name + "#split"] = (self._to_dev(pack_conv_weight(np.concatenate([w, w], axis=3))),) + self._dev[name][1:]
        self.input = torch.zeros((self.B, self.H, self.W, 3), dtype=torch.uint8, device=self.device)
        self.pred = torch.zeros((self.B, self.P, self.C), dtype=torch.float32, device=self.device)
        self.ops = []       # (_lib.PlanOp)
        self.op_info = []   # dict(name, flops, bytes)
        self._hilo = {}     # (hi, lo) pairs of the backbone taps (mixed precision, split lateral layers)
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
        self.ablated = bool(abl)
        if abl and os.environ.get("OD_ALLOW_ABLATION") != "1":
            # a leaked variable must not silently drop layers from a product plan: the dev scripts set both
            raise _lib.OdError("OD_ABLATE_OPS is set (a TIMING ablation: it drops layers, results are garbage) without "
                               "OD_ALLOW_ABLATION=1; unset it")
        if abl:
            import warnings
            warnings.warn(f"OD_ABLATE_OPS={os.environ['OD_ABLATE_OPS']!r}: layers dropped from the plan, outputs are garbage")
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

    def _buf32(self, h, w, c):
        t = torch.empty((self.B, h, w, c), dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def _wide(self, name, y32, res, res_f32, out32, out16, hilo, h, w, c, up2=False):
        """od_wide_add as a plan op: v = y32 (+ res, or + its nearest-neighbour parent on the half-size map `res` with
        up2); out32 = v, out16 = f16(v), hilo = [f16(v) | f16(v - f16(v))]."""
        d = _lib.WideDesc()
        d.y = y32.data_ptr()
        d.res = res.data_ptr() if res is not None else None
        d.out32 = out32.data_ptr() if out32 is not None else None
        d.out16 = out16.data_ptr() if out16 is not None else None
        d.out_hilo = hilo.data_ptr() if hilo is not None else None
        d.M, d.C, d.res_f32 = self.B * h * w, c, int(bool(res_f32))
        d.res_up2, d.H, d.W = int(bool(up2)), h, w
        op = _lib.PlanOp()
        op.kind = _lib.OD_OP_WIDE
        op.wide = d
        self.ops.append(op)
        m = self.B * h * w
        nbytes = m * c * (4 + (0 if res is None else (4 if res_f32 else 2)) + (4 if out32 is not None else 0)
                          + (2 if out16 is not None else 0) + (4 if hilo is not None else 0))
        self.op_info.append(dict(name=name, flops=0.0, bytes=float(nbytes), shape=(m, c, 0), kind="wide", y=y32, res=res,
                                 res_f32=bool(res_f32), out32=out32, out16=out16, hilo=hilo))

    def _neck_conv(self, name, x, x_hilo, h, w, cin, cout, k, act, up16=None, up32=None, need16=False, keep32=False,
                   consumers=None):
        """One neck / prediction-module layer of either plan -> (out f16 or None, out [hi | lo] or None, out f32 or None).
        The layer READS the (hi, lo) pair when it is a split layer.  It WRITES a pair (f32 conv output + od_wide_add) when
        one of its consumers is a split layer, and a plain f16 tensor when a consumer is not (or need16).  up16 / up32: the
        half-size map of an FPN sum -- f16 in the conv's own epilogue (default plan), or f32 in od_wide_add (mixed plan:
        the sum is formed in f32 and rounded once).  keep32: the f32 output is itself the operand of a later f32 sum."""
        consumers = _NECK_CONSUMERS.get(name, ()) if consumers is None else consumers
        want_hilo = any(c in self.split for c in consumers)
        want16 = need16 or any(c not in self.split for c in consumers)
        src, cin_eff, wkey = x, cin, None
        if name in self.split:
            if x_hilo is None:
                raise ValueError(f"split layer {name}: its input has no (hi, lo) pair")
            src, cin_eff, wkey = x_hilo, 2 * cin, name + "#split"
        if not want_hilo and up32 is None and not keep32:
            out, _, _ = self._conv(name, src, h, w, cin_eff, cout, k, 1, act, res=up16,
                                   res_mode=_lib.OD_RES_UP2 if up16 is not None else _lib.OD_RES_NONE, wkey=wkey)
            return out, None, None
        y32 = self._buf32(h, w, cout)
        in_epilogue = up16 if up32 is None else None
        self._conv(name, src, h, w, cin_eff, cout, k, 1, act, res=in_epilogue,
                   res_mode=_lib.OD_RES_UP2 if in_epilogue is not None else _lib.OD_RES_NONE, out=y32, out_f32=True, wkey=wkey)
        out16 = self._buf(h, w, cout) if want16 else None
        hilo = self._buf(h, w, 2 * cout) if want_hilo else None
        s32 = self._buf32(h, w, cout) if (keep32 and up32 is not None) else None
        if out16 is not None or hilo is not None or s32 is not None:
            self._wide(name + ".wide", y32, up32, True, s32, out16, hilo, h, w, cout, up2=up32 is not None)
        return out16, hilo, (s32 if s32 is not None else y32)

    def _conv(self, name, x, h, w, cin, cout, k, stride, act, res=None, res_mode=_lib.OD_RES_NONE, out=None,
              out_f32=False, obs=0, ops=0, then=None, wkey=None):
        """then = (name2, cout2, act2): the pointwise layer that consumes this layer's output rides in the same op
        (od_conv_desc.w2: inside the 8-wave kernel's epilogue, or as a second launch of the library's choosing);
        returns (out, ho, wo, out2) then."""
        wt, sc, bi = self._dev[wkey or name]
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
        if wkey is not None and wkey.endswith("#split"):  # algorithmic flops: the (hi, lo) pair is ONE operand
            self.op_info[-1]["flops"] *= 0.5
            self.op_info[-1]["split"] = True
        if then is not None:
            inf = self.op_info[-1]
            inf["flops"] += 2.0 * m * cout2 * cout
            inf["bytes"] += float(m * cout2 * 2 + cout2 * cout * 2)  # the 1x1's input never comes back from HBM
            inf["then"] = dict(name=name2, out=out2, act=act2)
            return out, ho, wo, out2
        return out, ho, wo

    def _conv_grouped(self, name, xs, dims, cin, cout, act, outs=None, out_f32=False, obs=0, ops=0):
        """One 3x3 stride-1 layer over several maps in ONE plan op (od_conv_desc.nseg): the prediction module shared by the
        pyramid levels.  xs: input tensors, dims: [(h, w)], outs: raw output pointers (else dense f16 buffers are made)."""
        wt, sc, bi = self._dev[name]
        if outs is None:
            bufs = [self._buf(h, w, cout) for h, w in dims]
            outs = [t.data_ptr() for t in bufs]
        else:
            bufs = [None] * len(dims)
        d = _lib.ConvDesc()
        d.w, d.scale, d.bias = wt.data_ptr(), sc.data_ptr(), bi.data_ptr()
        d.B, d.Cin, d.Cout, d.ksize, d.stride = self.B, cin, cout, 3, 1
        d.act, d.alpha = _lib.ACT_ENUM[act[0] if act else None], float(act[1]) if act else 0.0
        d.out_dtype = _lib.OD_DT_F32 if out_f32 else _lib.OD_DT_F16
        d.out_batch_stride, d.out_pix_stride = obs, ops
        d.tile_cfg = self.tile_cfg.get(name, self.auto_cfg)
        d.nseg = len(xs)
        for i, (x, (h, w), o) in enumerate(zip(xs, dims, outs)):
            d.seg_x[i], d.seg_out[i], d.seg_H[i], d.seg_W[i] = x.data_ptr(), o, h, w
        op = _lib.PlanOp()
        op.kind = _lib.OD_OP_CONV
        op.conv = d
        self.ops.append(op)
        m = sum(self.B * h * w for h, w in dims)
        if self.splitk and max(self.B * h * w for h, w in dims) * cout <= (1 << 22):
            # small maps (batch 1): the library runs the segments as separate launches and may split K for them
            self._splitk_elems = max(self._splitk_elems, max(self.B * h * w for h, w in dims) * cout)
            self._splitk_descs.append(len(self.ops) - 1)
        osz = 4 if out_f32 else 2
        self.op_info.append(dict(name=name, flops=2.0 * m * cout * 9 * cin,
                                 bytes=float(m * cin * 2 + m * cout * osz + cout * 9 * cin * 2), shape=(m, cout, 9 * cin),
                                 kind="conv_group", xs=list(xs), dims=list(dims), outs=bufs, act=act, out_f32=out_f32))
        return bufs

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
            wide = si in self.stream_stages  # f32 residual stream (precision="mixed")
            if wide:
                ride = False  # a block's 1x1 reads the f16 copy that od_wide_add rounds from the f32 sum
                hilo_tap = f"n.lat{si}" in self.split
                x32 = y32 = None
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
                if wide:
                    # y32 = act(bn(conv3x3(t))) as f32; x32 (+)= y32; x = f16(x32) (rounded once from the f32 sum)
                    if y32 is None:
                        y32 = self._buf32(h, w, ch)
                        x32 = self._buf32(h, w, ch)
                    self._conv(f"b.s{si}.{r}.b", t, h, w, ch // 2, ch, 3, 1, bact, out=y32, out_f32=True)
                    last = r + 1 == n
                    xn = self._buf(h, w, ch)
                    hl = self._buf(h, w, 2 * ch) if (last and hilo_tap) else None
                    self._wide(f"b.s{si}.{r}.add", y32, x if r == 0 else x32, r != 0, x32, xn, hl, h, w, ch)
                    x = xn
                    if hl is not None:
                        self._hilo[f"tap{si}"] = hl
                    t = None
                elif ride and r + 1 < n:
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
        # FPN sums (docs/MODEL.md:5-8): in the mixed plan they are formed in f32 (od_wide_add adds the f32 half-size map)
        wf = self.precision == "mixed" and self.wide_fpn
        p5, p5s, p5w = self._neck_conv("n.lat5", c5, self._hilo.get("tap5"), h5, w5, ch5, nc, 1, hact, need16=not wf, keep32=wf)
        m4, m4s, _ = self._neck_conv("n.lat4", c4, self._hilo.get("tap4"), h4, w4, ch4, nc, 1, hact,
                                     up16=None if wf else p5, up32=p5w if wf else None)
        p4, p4s, p4w = self._neck_conv("n.out4", m4, m4s, h4, w4, nc, nc, 3, hact, need16=not wf, keep32=wf)
        m3, m3s, _ = self._neck_conv("n.lat3", c3, self._hilo.get("tap3"), h3, w3, ch3, nc, 1, hact,
                                     up16=None if wf else p4, up32=p4w if wf else None)
        p3, p3s, _ = self._neck_conv("n.out3", m3, m3s, h3, w3, nc, nc, 3, hact)
        self.levels = [(p3, h3, w3), (p4, h4, w4), (p5, h5, w5)]
        self.taps = [c3, c4, c5]
        # shared prediction module; the last conv writes f32 logits into pred[:, off:off+h*w*8, :]
        off = 0
        cout = W.NUM_PRIORS * self.C
        group = (self.precision == "f16" and self.tower == 1 and nc % 64 == 0
                 and os.environ.get("OD_GROUP_HEAD", "1") != "0")
        if group:
            # the prediction module's weights are shared by the three levels (docs/MODEL.md:8): ONE launch per layer over
            # all 67 200 x B / 8 rows instead of three (the 20^2 and 10^2 launches filled 50 and 13 of 256 CUs)
            dims = [(h, w) for _x, h, w in self.levels]
            ts = self._conv_grouped("h.t0", [x for x, _h, _w in self.levels], dims, nc, nc, hact)
            ptrs, rows = [], []
            for h, w in dims:
                ptrs.append(self.pred.data_ptr() + off * self.C * 4)
                rows.append((off, h * w * W.NUM_PRIORS))
                off += h * w * W.NUM_PRIORS
            self._conv_grouped("h.out", ts, dims, nc, cout, None, outs=ptrs, out_f32=True, obs=self.P * self.C, ops=cout)
            self.op_info[-1]["pred_rows"] = rows
        for (x, h, w), xs in ([] if group else zip(self.levels, (p3s, p4s, p5s))):
            t, ts = x, xs
            for i in range(self.tower):
                t, ts, _ = self._neck_conv(f"h.t{i}", t, ts, h, w, nc, nc, 3, hact,
                                           consumers=(f"h.t{i + 1}",) if i + 1 < self.tower else ("h.out",))
            out_ptr = self.pred.data_ptr() + off * self.C * 4
            if "h.out" in self.split:
                self._conv("h.out", ts, h, w, 2 * nc, cout, 3, 1, None, out=out_ptr, out_f32=True,
                           obs=self.P * self.C, ops=cout, wkey="h.out#split")
            else:
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
