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
