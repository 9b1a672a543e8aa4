"""GPU parity: od_stem_fwd (uint8 image -> conv 3->32 -> stride-2 conv 32->64 in one launch) vs the CPU oracle and vs the
two-kernel HIP path (od_conv_first_fwd + od_conv2d_fwd).  Same rounding points everywhere: the 32-channel tensor is
rounded to f16 once, the output once; tolerance as in test_gpu_conv.py."""
import numpy as np
import pytest
import torch

from oracle import network as onet

pytestmark = pytest.mark.gpu


def _leaky(y, a):
    return np.where(y > 0, y, y * a)


def _inputs(B, H, W, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    w0 = (rng.normal(0, 1, (32, 3, 3, 3)) * np.sqrt(2.0 / 27)).astype(np.float16).astype(np.float32)
    w3 = (rng.normal(0, 1, (64, 3, 3, 32)) * np.sqrt(2.0 / 288)).astype(np.float16).astype(np.float32)
    s0 = (rng.uniform(0.5, 1.5, 32) / 255.0).astype(np.float32)
    b0 = rng.normal(0, 0.1, 32).astype(np.float32)
    s3 = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    b3 = rng.normal(0, 0.1, 64).astype(np.float32)
    return x, w0, s0, b0, w3, s3, b3


def _actf(y, act, a):
    if act == "leaky":
        return _leaky(y, a)
    if act == "elu":
        return np.where(y > 0, y, a * np.expm1(np.minimum(y, 0)))
    return y


@pytest.mark.parametrize("act", ["leaky", "elu", None], ids=str)  # each activation is its own compiled kernel
@pytest.mark.parametrize("shape", [(1, 32, 32), (2, 64, 96), (3, 96, 32), (1, 160, 64)], ids=str)
def test_stem_matches_oracle(cuda, shape, act):
    from object_detector_amd import ops
    B, H, W = shape
    a = 0.1 if act == "leaky" else 1.0
    x, w0, s0, b0, w3, s3, b3 = _inputs(B, H, W, hash(shape) & 0xFFFF)
    out = ops.stem(torch.from_numpy(x).to(cuda), w0, s0, b0, w3, s3, b3, act=act, alpha=a)
    got = out.cpu().numpy().astype(np.float64)
    t = onet.conv_nhwc(x.astype(np.float32), w0, 1, torch.float64).astype(np.float64) * s0 + b0
    t = _actf(t, act, a).astype(np.float16).astype(np.float32)
    ref = _actf(onet.conv_nhwc(t, w3, 2, torch.float64).astype(np.float64) * s3 + b3, act, a)
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


def test_stem_equals_two_kernel_path(cuda):
    from object_detector_amd import ops
    x, w0, s0, b0, w3, s3, b3 = _inputs(2, 64, 64, 3)
    xt = torch.from_numpy(x).to(cuda)
    t = ops.conv_first(xt, w0, s0, b0, "leaky", 0.1)
    two = ops.conv2d(t, w3, s3, b3, stride=2, act="leaky", alpha=0.1)
    one = ops.stem(xt, w0, s0, b0, w3, s3, b3, act="leaky", alpha=0.1)
    torch.cuda.synchronize()
    a, b = one.float().cpu().numpy(), two.float().cpu().numpy()
    assert np.abs(a - b).max() <= 2.0 ** -9 * max(1.0, np.abs(b).max())


def test_stem_rejects_unsupported(cuda):
    from object_detector_amd import ops, _lib
    x = torch.zeros((1, 48, 32, 3), dtype=torch.uint8, device=cuda)  # H not a multiple of 32
    _x, w0, s0, b0, w3, s3, b3 = _inputs(1, 32, 32, 0)
    with pytest.raises(_lib.OdError):
        ops.stem(x, w0, s0, b0, w3, s3, b3)
