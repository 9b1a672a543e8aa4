"""Parameter container of the detector: layer list, random initialisation, (de)serialisation.

The network follows reference docs/MODEL.md:5-27 (Darknet53 base, FPN-like neck, prediction module shared by the
three feature maps, 8 prior boxes per cell).  The reference stores trained weights as Keras `*.h5` (LFS,
.gitattributes:5) which are not in the tree; this build's own format is a flat `.npz` / safetensors of
    <layer>.w      f32 [Cout, k, k, Cin]   (OHWI)
    <layer>.gamma/.beta/.mean/.var  f32 [Cout]   (BatchNorm, eps 1e-3)   or   <layer>.bias  (prediction conv)
`load_voc` reads it from a LOCAL path only.
"""
from __future__ import annotations

import pathlib

import numpy as np

BN_EPS = 1e-3
STAGES = ((1, 64), (2, 128), (8, 256), (8, 512), (4, 1024))  # Darknet53: (residual blocks, channels) per stage
NUM_PRIORS = 8


def layer_specs(num_classes=20, neck_ch=256, tower=1):
    """[(name, cin, cout, ksize, stride, has_bn)] in execution order."""
    specs = [("b.conv0", 3, 32, 3, 1, True)]
    cin = 32
    for si, (n, ch) in enumerate(STAGES, start=1):
        specs.append((f"b.down{si}", cin, ch, 3, 2, True))
        for r in range(n):
            specs.append((f"b.s{si}.{r}.a", ch, ch // 2, 1, 1, True))
            specs.append((f"b.s{si}.{r}.b", ch // 2, ch, 3, 1, True))
        cin = ch
    specs += [("n.lat5", 1024, neck_ch, 1, 1, True), ("n.lat4", 512, neck_ch, 1, 1, True),
              ("n.out4", neck_ch, neck_ch, 3, 1, True), ("n.lat3", 256, neck_ch, 1, 1, True),
              ("n.out3", neck_ch, neck_ch, 3, 1, True)]
    specs += [(f"h.t{t}", neck_ch, neck_ch, 3, 1, True) for t in range(tower)]
    specs.append(("h.out", neck_ch, NUM_PRIORS * (2 + num_classes + 4), 3, 1, False))
    return specs


def random_init(seed=2, num_classes=20, neck_ch=256, tower=1):
    """Synthetic parameters for benchmarking / parity tests (no trained weights exist offline)."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, cin, cout, k, _s, bn in layer_specs(num_classes, neck_ch, tower):
        std = np.sqrt(2.0 / (cin * k * k)) if bn else 0.01
        w = rng.normal(0.0, std, size=(cout, k, k, cin)).astype(np.float32)
        params[name + ".w"] = w.astype(np.float16).astype(np.float32)
        if bn:
            lo, hi = (0.1, 0.3) if (name.startswith("b.s") and name.endswith(".b")) else (0.5, 1.5)
            params[name + ".gamma"] = rng.uniform(lo, hi, cout).astype(np.float32)
            params[name + ".beta"] = rng.normal(0, 0.1, cout).astype(np.float32)
            params[name + ".mean"] = rng.normal(0, 0.1, cout).astype(np.float32)
            params[name + ".var"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        else:
            params[name + ".bias"] = rng.normal(0, 0.1, cout).astype(np.float32)
    return params


def fold_bn(params, name):
    """Inference epilogue constants: y = scale * conv + bias."""
    if name + ".gamma" in params:
        scale = params[name + ".gamma"] / np.sqrt(params[name + ".var"] + np.float32(BN_EPS))
        bias = params[name + ".beta"] - params[name + ".mean"] * scale
        return scale.astype(np.float32), bias.astype(np.float32)
    cout = params[name + ".w"].shape[0]
    return np.ones(cout, np.float32), params[name + ".bias"].astype(np.float32)


def infer_arch(params):
    """(num_classes, neck_ch, tower) from the tensor shapes."""
    neck_ch = params["n.lat5.w"].shape[0]
    tower = sum(1 for k in params if k.startswith("h.t") and k.endswith(".w"))
    num_classes = params["h.out.w"].shape[0] // NUM_PRIORS - 6
    return num_classes, neck_ch, tower


def save(path, params, meta=None):
    path = pathlib.Path(path)
    extra = {f"__meta__.{k}": np.asarray(v) for k, v in (meta or {}).items()}
    np.savez(path, **params, **extra)


def load(path):
    """-> (params, meta).  numpy .npz, allow_pickle=False."""
    with np.load(pathlib.Path(path), allow_pickle=False) as z:
        params = {k: z[k] for k in z.files if not k.startswith("__meta__.")}
        meta = {k[len("__meta__."):]: z[k] for k in z.files if k.startswith("__meta__.")}
    return params, meta


# ---- Darknet backbone import --------------------------------------------------------------
# The reference's base network is "Darknet53 (YOLOv3's base network)" (docs/MODEL.md:15-17); its public pre-trained file is
# `darknet53.conv.74`.  Layout of a Darknet weights file [published format, pjreddie/darknet `parser.c::load_weights`]:
#   int32 major, minor, revision; then `seen` as int64 if major*10 + minor >= 2 else int32;
#   then per convolutional layer, in network order, float32 little endian:
#       batch-normalised layer:  beta[Cout], gamma[Cout], running_mean[Cout], running_var[Cout], w[Cout][Cin][k][k]
#       plain layer:             bias[Cout], w[Cout][Cin][k][k]
# Darknet's BatchNorm epsilon is 1e-6 (blas.c::normalize_cpu uses sqrt(var) + .000001f); this build folds BN with
# BN_EPS, so the import rewrites var so that gamma / sqrt(var' + BN_EPS) equals Darknet's gamma / (sqrt(var) + 1e-6).
DARKNET_BN_EPS = 1e-6


def backbone_specs():
    return [s for s in layer_specs() if s[0].startswith("b.")]


def load_darknet_backbone(path, params=None):
    """Read the 52 backbone convolutions of a Darknet `darknet53.conv.74`-style file (a LOCAL file, never downloaded) into
    `params` (a fresh random-init detector if None).  Returns (params, n_floats_read).  Raises ValueError when the
    file is shorter than the backbone needs."""
    raw = np.fromfile(pathlib.Path(path), dtype=np.uint8)
    if raw.size < 16:
        raise ValueError(f"{path}: not a Darknet weights file (too short)")
    major, minor, _rev = np.frombuffer(raw[:12].tobytes(), dtype="<i4")
    off = 12 + (8 if int(major) * 10 + int(minor) >= 2 else 4)
    data = np.frombuffer(raw[off:off + (raw.size - off) // 4 * 4].tobytes(), dtype="<f4")
    params = dict(random_init() if params is None else params)
    pos = 0

    def take(n):
        nonlocal pos
        if pos + n > data.size:
            raise ValueError(f"{path}: ends after {data.size} floats; the Darknet53 backbone needs more")
        out = data[pos:pos + n]
        pos += n
        return out

    for name, cin, cout, k, _s, _bn in backbone_specs():
        beta, gamma, mean, var = (take(cout).astype(np.float32) for _ in range(4))
        w = take(cout * cin * k * k).reshape(cout, cin, k, k)
        params[name + ".w"] = np.ascontiguousarray(w.transpose(0, 2, 3, 1)).astype(np.float32)  # OIHW -> OHWI
        params[name + ".gamma"], params[name + ".beta"], params[name + ".mean"] = gamma, beta, mean
        sd = np.sqrt(np.maximum(var, 0.0)) + np.float32(DARKNET_BN_EPS)
        params[name + ".var"] = (sd * sd - np.float32(BN_EPS)).astype(np.float32)  # may be < 0: only var + BN_EPS is used
    return params, pos


def save_darknet_backbone(path, params, major=0, minor=2, revision=0, seen=0):
    """Inverse of load_darknet_backbone (round-trip tests; export for Darknet-side comparison)."""
    chunks = [np.asarray([major, minor, revision], "<i4").tobytes(),
              np.asarray([seen], "<i8" if major * 10 + minor >= 2 else "<i4").tobytes()]
    for name, _cin, _cout, _k, _s, _bn in backbone_specs():
        sd = np.sqrt(params[name + ".var"].astype(np.float64) + BN_EPS) - DARKNET_BN_EPS
        for v in (params[name + ".beta"], params[name + ".gamma"], params[name + ".mean"], (sd * sd)):
            chunks.append(np.asarray(v, "<f4").tobytes())
        chunks.append(np.ascontiguousarray(params[name + ".w"].transpose(0, 3, 1, 2)).astype("<f4").tobytes())
    pathlib.Path(path).write_bytes(b"".join(chunks))
