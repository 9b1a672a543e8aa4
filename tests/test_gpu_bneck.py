"""GPU parity: od_bottleneck_fwd (fused 1x1 -> 3x3 -> +x residual block) vs the CPU oracle.

The oracle composes the two convolutions with the SAME rounding points as the kernel (and as the two-layer HIP path):
the middle tensor is rounded to f16 once, the output is rounded to f16 once; everything else is f64.  Tolerance as in
test_gpu_conv.py: 1 f16 ulp + 1e-3 of the output scale (the middle rounding can flip by one ulp on accumulation-order
noise, which the 3x3 then averages over 9*C/2 terms).
"""
import numpy as np
import pytest
import torch

from oracle import network as onet

pytestmark = pytest.mark.gpu


def _act(y, act, alpha):
    if act == "leaky":
        return np.where(y > 0, y, y * alpha)
    if act == "elu":
        return np.where(y > 0, y, alpha * np.expm1(np.minimum(y, 0)))
    return y


def _ref(x, w1, s1, b1, w3, s3, b3, act, alpha):
    t = onet.conv_nhwc(x, w1, 1, torch.float64).astype(np.float64) * s1 + b1
    t = _act(t, act, alpha).astype(np.float16).astype(np.float32)  # the one rounding of the middle tensor
    y = onet.conv_nhwc(t, w3, 1, torch.float64).astype(np.float64) * s3 + b3
    return _act(y, act, alpha) + x.astype(np.float64)


CASES = [
    # B, H, W, C, act
    (1, 16, 16, 64, "leaky"),     # one tile: every border is an image border
    (2, 32, 48, 64, "leaky"),     # interior halos between tiles, non-square
    (1, 16, 32, 128, "leaky"),    # streamed 3x3 weights
    (2, 48, 32, 128, "leaky"),
    (1, 32, 32, 64, "elu"),
    (1, 32, 16, 128, None),
    # every (C, activation) pair is its own compiled kernel since round 2 (the activation is a template parameter)
    (2, 32, 32, 64, None),
    (1, 16, 32, 128, "elu"),
    (33, 32, 32, 64, "leaky"),    # 132 tiles... C = 64 is persistent: more tiles than one round of its loader / compute waves at small grids
]


@pytest.mark.parametrize("case", CASES, ids=str)
def test_bottleneck_matches_oracle(cuda, case):
    from object_detector_amd import ops
    B, H, W, C, act = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = rng.normal(0, 1, (B, H, W, C)).astype(np.float16)
    w1 = (rng.normal(0, 1, (C // 2, 1, 1, C)) * np.sqrt(2.0 / C)).astype(np.float16)
    w3 = (rng.normal(0, 1, (C, 3, 3, C // 2)) * np.sqrt(2.0 / (9 * C // 2))).astype(np.float16)
    s1 = rng.uniform(0.5, 1.5, C // 2).astype(np.float32)
    b1 = rng.normal(0, 0.1, C // 2).astype(np.float32)
    s3 = rng.uniform(0.5, 1.5, C).astype(np.float32)
    b3 = rng.normal(0, 0.1, C).astype(np.float32)
    alpha = 0.1 if act == "leaky" else 1.0
    out = ops.bottleneck(torch.from_numpy(x).to(cuda), w1.astype(np.float32), s1, b1, w3.astype(np.float32), s3, b3,
                         act=act, alpha=alpha)
    got = out.cpu().numpy().astype(np.float64)
    ref = _ref(x.astype(np.float32), w1.astype(np.float32), s1, b1, w3.astype(np.float32), s3, b3, act, alpha)
    err = np.abs(got - ref)
    tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


def test_bottleneck_equals_two_layer_path(cuda):
    """Same inputs through od_conv2d_fwd x 2 (+ residual) and through the fused kernel: identical rounding points, so the
    outputs agree to one f16 ulp of the output."""
    from object_detector_amd import ops
    rng = np.random.default_rng(5)
    B, H, W, C = 2, 32, 32, 128
    x = rng.normal(0, 1, (B, H, W, C)).astype(np.float16)
    w1 = (rng.normal(0, 1, (C // 2, 1, 1, C)) * np.sqrt(2.0 / C)).astype(np.float32)
    w3 = (rng.normal(0, 1, (C, 3, 3, C // 2)) * np.sqrt(2.0 / (9 * C // 2))).astype(np.float32)
    s1, b1 = rng.uniform(0.5, 1.5, C // 2).astype(np.float32), rng.normal(0, 0.1, C // 2).astype(np.float32)
    s3, b3 = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 0.1, C).astype(np.float32)
    xt = torch.from_numpy(x).to(cuda)
    t = ops.conv2d(xt, w1, s1, b1, act="leaky", alpha=0.1)
    two = ops.conv2d(t, w3, s3, b3, act="leaky", alpha=0.1, res=xt, res_mode="same")
    one = ops.bottleneck(xt, w1, s1, b1, w3, s3, b3, act="leaky", alpha=0.1)
    torch.cuda.synchronize()
    a, b = one.float().cpu().numpy(), two.float().cpu().numpy()
    assert np.abs(a - b).max() <= 2.0 ** -9 * max(1.0, np.abs(b).max())


def test_bottleneck_rejects_unsupported(cuda):
    from object_detector_amd import ops, _lib
    x = torch.zeros((1, 16, 16, 256), dtype=torch.float16, device=cuda)
    with pytest.raises(_lib.OdError):
        ops.bottleneck(x, np.zeros((128, 1, 1, 256), np.float32), np.ones(128), np.zeros(128),
                       np.zeros((256, 3, 3, 128), np.float32), np.ones(256), np.zeros(256))
