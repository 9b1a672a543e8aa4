"""Oracle forward pass: Darknet53 backbone + FPN-like neck + shared prediction module (numpy / torch-CPU, NHWC).

Follows reference docs/MODEL.md:
  :5-8   "shrink then enlarge again, FPN-like; prediction-module weights are shared"
  :15-17 base network = Darknet53 (YOLOv3's), i.e. conv -> BN -> LeakyReLU(0.1), stages of [1x1 C->C/2, 3x3 C/2->C, +x]
         repeated 1,2,8,8,4 times after stride-2 3x3 downsamples (YOLOv3 paper, table 1)
  :19-21 ELU in the added layers
  :23-27 three feature maps (stride 8/16/32), 8 prior boxes per cell
and the target/prediction row layout of reference check_assign.py:25-27: [not-obj, obj, NC classes, 4 box offsets].

[BUILD-DEFINED] (the reference tree does not specify them): neck width 256, one 3x3 tower conv in the prediction module,
BatchNorm eps 1e-3 (Keras default), level order (stride 8, 16, 32), row order inside a level (y, x, prior).

Two arithmetic modes:
  storage="f32": activations stay f32 between layers (the "plain fp32 reference")
  storage="f16": activations are rounded to f16 wherever the HIP path stores them (after every fused conv epilogue),
                 weights are the same f16-rounded values in both modes; accumulation is f32/f64 on the CPU
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
STAGES = ((1, 64), (2, 128), (8, 256), (8, 512), (4, 1024))  # (residual blocks, channels)
NECK_CH = 256
NUM_PRIORS = 8


def layer_specs(num_classes: int = 20, neck_ch: int = NECK_CH, tower: int = 1):
    """[(name, cin, cout, ksize, stride, has_bn)] for every conv, in execution order of the backbone then neck/head."""
    specs = [("b.conv0", 3, 32, 3, 1, True)]
    cin = 32
    for si, (n, ch) in enumerate(STAGES, start=1):
        specs.append((f"b.down{si}", cin, ch, 3, 2, True))
        for r in range(n):
            specs.append((f"b.s{si}.{r}.a", ch, ch // 2, 1, 1, True))
            specs.append((f"b.s{si}.{r}.b", ch // 2, ch, 3, 1, True))
        cin = ch
    specs += [
        ("n.lat5", 1024, neck_ch, 1, 1, True),
        ("n.lat4", 512, neck_ch, 1, 1, True),
        ("n.out4", neck_ch, neck_ch, 3, 1, True),
        ("n.lat3", 256, neck_ch, 1, 1, True),
        ("n.out3", neck_ch, neck_ch, 3, 1, True),
    ]
    for t in range(tower):
        specs.append((f"h.t{t}", neck_ch, neck_ch, 3, 1, True))
    specs.append(("h.out", neck_ch, NUM_PRIORS * (2 + num_classes + 4), 3, 1, False))
    return specs


def init_weights(seed: int = 2, num_classes: int = 20, neck_ch: int = NECK_CH, tower: int = 1):
    """Random-init parameters (SURVEY.md §8d recipe): He-normal conv, BN gamma~U(.5,1.5), beta~N(0,.1),
    mean~N(0,.1), var~U(.5,1.5); weights are stored OHWI f32 but already rounded to f16-representable values."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, cin, cout, k, _s, bn in layer_specs(num_classes, neck_ch, tower):
        fan_in = cin * k * k
        std = np.sqrt(2.0 / fan_in) if bn else 0.01
        w = rng.normal(0.0, std, size=(cout, k, k, cin)).astype(np.float32)
        params[name + ".w"] = w.astype(np.float16).astype(np.float32)
        if bn:
            # residual-branch outputs (b.sX.Y.b) get a small gamma so 23 stacked shortcuts do not blow up the
            # activation scale with random (un-trained) statistics; everything else gamma~U(.5,1.5)
            lo, hi = (0.1, 0.3) if (name.startswith("b.s") and name.endswith(".b")) else (0.5, 1.5)
            params[name + ".gamma"] = rng.uniform(lo, hi, cout).astype(np.float32)
            params[name + ".beta"] = rng.normal(0, 0.1, cout).astype(np.float32)
            params[name + ".mean"] = rng.normal(0, 0.1, cout).astype(np.float32)
            params[name + ".var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        else:
            params[name + ".bias"] = rng.normal(0, 0.1, cout).astype(np.float32)
    return params


def f16_weights(params):
    """The model the device runs: conv weights rounded to f16 (what od_pack_weights stores), everything else f32.
    init_weights() already returns f16-representable weights; TRAINED masters are f32 and need this before a comparison
    that is about arithmetic, not about weight quantisation."""
    return {k: (np.asarray(v, np.float32).astype(np.float16).astype(np.float32) if k.endswith(".w") else v)
            for k, v in params.items()}


def fold_bn(params, name):
    """(scale, bias) f32 of the fused epilogue: BN(x) = gamma*(x-mean)/sqrt(var+eps)+beta."""
    if name + ".gamma" in params:
        scale = params[name + ".gamma"] / np.sqrt(params[name + ".var"] + np.float32(BN_EPS))
        bias = params[name + ".beta"] - params[name + ".mean"] * scale
        return scale.astype(np.float32), bias.astype(np.float32)
    cout = params[name + ".w"].shape[0]
    return np.ones(cout, np.float32), params[name + ".bias"].astype(np.float32)


def _act(y, act, alpha):
    if act == "leaky":
        return np.where(y > 0, y, y * np.float32(alpha)).astype(y.dtype)
    if act == "elu":
        return np.where(y > 0, y, np.float32(alpha) * np.expm1(np.minimum(y, 0))).astype(y.dtype)
    return y


def conv_nhwc(x, w_ohwi, stride, dtype=torch.float32):
    """'same'-padded conv, x [B,H,W,Cin] f32, w [Cout,k,k,Cin]; torch CPU conv2d (oneDNN) in `dtype`."""
    k = w_ohwi.shape[1]
    xt = torch.from_numpy(np.ascontiguousarray(x)).to(dtype).permute(0, 3, 1, 2)
    wt = torch.from_numpy(np.ascontiguousarray(w_ohwi)).to(dtype).permute(0, 3, 1, 2)
    y = F.conv2d(xt, wt, stride=stride, padding=k // 2)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def conv_nhwc_numpy(x, w_ohwi, stride):
    """Independent slow restatement (explicit taps, einsum in f64) used only to cross-check conv_nhwc on small cases."""
    B, H, W, Cin = x.shape
    Cout, k, _, _ = w_ohwi.shape
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, Cin), np.float64)
    xp[:, pad:pad + H, pad:pad + W] = x
    out = np.zeros((B, Ho, Wo, Cout), np.float64)
    for dy in range(k):
        for dx in range(k):
            patch = xp[:, dy:dy + (Ho - 1) * stride + 1:stride, dx:dx + (Wo - 1) * stride + 1:stride]
            out += np.einsum("bhwc,oc->bhwo", patch, w_ohwi[:, dy, dx].astype(np.float64))
    return out


class MixedPlan:
    """The three predicates that restate object_detector_amd.net's precision="mixed" plan for Runner:
      storage(name)     -> True: that layer's OUTPUT is rounded to f16 where it is stored
      operand_f16(name) -> True: that layer's INPUT is rounded to f16 before the multiply (every layer but the split ones,
                           whose (hi, lo) operand pair carries ~22 bits: modelled as the unrounded f32 value)
      res_f16(name)     -> True: the residual operand is the f16 copy (the FPN up2 adds); False: the f32 residual stream
    A residual-stream tensor of `stream_stages` is f32; the first block of a stage adds to the (f16) stride-2 conv output.
    A neck / head tensor is kept f32 iff one of its consumers is a split layer."""
    CONSUMERS = {"n.lat5": ("h.t0",), "n.lat4": ("n.out4",), "n.out4": ("h.t0",), "n.lat3": ("n.out3",), "n.out3": ("h.t0",),
                 "h.t0": ("h.out",)}

    def __init__(self, stream_stages=(4, 5), split=("n.lat4", "n.lat5", "n.out3", "n.out4", "h.t0", "h.out"), wide_fpn=True):
        """wide_fpn: the two FPN sums lat(c) + up2(p) are formed in f32 from the f32 half-size map (else the conv epilogue
        adds the f16 copy)."""
        self.stream_stages, self.split, self.wide_fpn = tuple(stream_stages), tuple(split), bool(wide_fpn)

    def _on_stream(self, name):
        return any(name.startswith(f"b.s{k}.") and name.endswith(".b") for k in self.stream_stages)

    def storage(self, name):
        if self._on_stream(name):
            return False
        if any(c in self.split for c in self.CONSUMERS.get(name, ())):
            return False
        if self.wide_fpn and name in ("n.lat5", "n.out4"):  # kept f32 as the operand of the next FPN sum
            return False
        return True

    def operand_f16(self, name):
        return name not in self.split

    def res_f16(self, name):
        if self.wide_fpn and name in ("n.lat4", "n.lat3"):
            return False
        return not self._on_stream(name)

    def runner(self, params, **kw):
        return Runner(params, storage=self.storage, operand_f16=self.operand_f16, res_f16=self.res_f16, **kw)


class Runner:
    """Layer-by-layer forward with the fused-epilogue semantics of od_conv2d_fwd (include/odhip.h)."""

    def __init__(self, params, storage="f16", precise=False, backbone_act=("leaky", 0.1), head_act=("elu", 1.0),
                 operand_f16=None, res_f16=None):
        """operand_f16: predicate layer name -> bool; True = that conv's INPUT is rounded to f16 before the multiply even when
        the tensor is stored wider (a wide residual stream whose convolutions still run on f16 MFMA operands); a layer for
        which it is False multiplies the stored values as they are (f32 storage + False = a hi/lo split-operand conv)."""
        self.p = params
        self.operand_f16 = operand_f16
        self.res_f16 = res_f16  # predicate: the residual operand of that layer is read as f16 (see MixedPlan)
        self.trace = None  # set to {} to record every layer's stored output: name -> [array per call] (debugging aid)
        self.storage = storage
        self.dtype = torch.float64 if precise else torch.float32
        self.backbone_act = backbone_act
        self.head_act = head_act

    def _store(self, y, name=None):
        """storage: "f32", "f16", or a predicate name -> bool choosing per layer OUTPUT whether it is rounded to f16 (error
        attribution: scripts/dev/attribute_logit_error.py; mixed-precision plans: object_detector_amd.net f32_layers)."""
        y = y.astype(np.float32)
        if self.storage == "f16" or (callable(self.storage) and self.storage(name)):
            return y.astype(np.float16).astype(np.float32)
        return y

    def conv(self, x, name, stride=1, act=None, res=None, res_up2=False, store=True):
        w = self.p[name + ".w"]
        scale, bias = fold_bn(self.p, name)
        if self.operand_f16 is not None and self.operand_f16(name):
            x = x.astype(np.float16).astype(np.float32)
        y = conv_nhwc(x, w, stride, self.dtype).astype(np.float32)
        y = y * scale + bias
        if act is not None:
            y = _act(y, act[0], act[1])
        if res is not None:
            r = res
            if self.res_f16 is not None and self.res_f16(name):
                r = r.astype(np.float16).astype(np.float32)
            if res_up2:
                r = np.repeat(np.repeat(r, 2, axis=1), 2, axis=2)
            y = y + r
        out = self._store(y, name) if store else y.astype(np.float32)
        if self.trace is not None:
            self.trace.setdefault(name, []).append(out)
        return out

    def first(self, x_u8, name="b.conv0"):
        w = self.p[name + ".w"]
        scale, bias = fold_bn(self.p, name)
        y = conv_nhwc(x_u8.astype(np.float32), w, 1, self.dtype).astype(np.float32)
        y = y * (scale / np.float32(255.0)).astype(np.float32) + bias
        return self._store(_act(y, *self.backbone_act), name)

    def backbone(self, x_u8):
        a = self.backbone_act
        x = self.first(x_u8)
        taps = []
        for si, (n, _ch) in enumerate(STAGES, start=1):
            x = self.conv(x, f"b.down{si}", stride=2, act=a)
            for r in range(n):
                h = self.conv(x, f"b.s{si}.{r}.a", act=a)
                x = self.conv(h, f"b.s{si}.{r}.b", act=a, res=x)
            taps.append(x)
        return taps[2], taps[3], taps[4]  # stride 8, 16, 32

    def neck(self, c3, c4, c5):
        a = self.head_act
        p5 = self.conv(c5, "n.lat5", act=a)
        m4 = self.conv(c4, "n.lat4", act=a, res=p5, res_up2=True)
        p4 = self.conv(m4, "n.out4", act=a)
        m3 = self.conv(c3, "n.lat3", act=a, res=p4, res_up2=True)
        p3 = self.conv(m3, "n.out3", act=a)
        return p3, p4, p5

    def head(self, levels, num_classes=20):
        a = self.head_act
        C = 2 + num_classes + 4
        outs = []
        tower = sum(1 for k in self.p if k.startswith("h.t") and k.endswith(".w"))
        for x in levels:
            t = x
            for i in range(tower):
                t = self.conv(t, f"h.t{i}", act=a)
            o = self.conv(t, "h.out", act=None, store=False)  # f32 logits, never rounded
            B, H, W, _ = o.shape
            outs.append(o.reshape(B, H * W * NUM_PRIORS, C))
        return np.concatenate(outs, axis=1)

    def forward(self, x_u8, num_classes=20):
        """uint8 [B,S,S,3] -> pred f32 [B,P,2+NC+4]"""
        return self.head(self.neck(*self.backbone(x_u8)), num_classes)


def synthetic_images(batch, size, seed=0):
    """VOC-shaped synthetic input: uint8 NHWC, i.i.d. uniform, default_rng(seed)."""
    return np.random.default_rng(seed).integers(0, 256, size=(batch, size, size, 3), dtype=np.uint8)
