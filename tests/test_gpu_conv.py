"""GPU parity: od_conv2d_fwd / od_conv_first_fwd (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerance: inputs and weights are f16-representable, the kernel accumulates in f32 and rounds ONCE to f16, so the
result must equal the f64 oracle rounded to f16 up to 1 f16 ulp (rtol 2^-10) + accumulation-order noise (atol 1e-3
relative to the output scale, the tolerance north_star states for logits).
"""
import numpy as np
import pytest
import torch

from oracle import network as onet

pytestmark = pytest.mark.gpu


def _ref(x, w, scale, bias, stride, act, alpha, res=None, up2=False):
    y = onet.conv_nhwc(x, w, stride, torch.float64).astype(np.float64)
    y = y * scale.astype(np.float64) + bias.astype(np.float64)
    if act == "leaky":
        y = np.where(y > 0, y, y * alpha)
    elif act == "elu":
        y = np.where(y > 0, y, alpha * np.expm1(np.minimum(y, 0)))
    if res is not None:
        r = res.astype(np.float64)
        if up2:
            r = np.repeat(np.repeat(r, 2, axis=1), 2, axis=2)
        y = y + r
    return y


NIG = 8               # implicit-GEMM table configs 0..7 (what pick_cfg can select: 0-3 generic, 4/5/7 wave-specialised, 6 deep ring)
SPEC = (4, 5, 7)      # the wave-specialised ones (tap-uniform only: Cin % 64 == 0)

CASES = [
    # B, H, W, Cin, Cout, k, stride, act, res, cfg
    (2, 16, 16, 32, 64, 3, 1, "leaky", "same", -1),     # Cin=32: per-lane tap path, K tail (288 -> 320)
    (2, 16, 16, 64, 128, 3, 2, "leaky", "none", -1),    # stride 2
    (3, 10, 10, 128, 64, 1, 1, "leaky", "none", -1),    # 1x1, M=300 (ragged M edge)
    (1, 8, 8, 64, 32, 1, 1, "leaky", "none", -1),       # Cout=32 < BN
    (2, 12, 20, 256, 256, 3, 1, "elu", "up2", -1),      # neck lateral-style, non-square
    (2, 10, 10, 512, 1024, 3, 1, "leaky", "same", -1),  # long K
    (1, 4, 4, 8, 8, 3, 1, None, "none", -1),            # tiny everything: Cin=8, single partial tile
    (2, 16, 16, 32, 64, 3, 2, "leaky", "none", 1),      # Cin=32: the generic (per-lane tap) 3x3 variant
    (2, 9, 9, 40, 72, 1, 1, "leaky", "none", 2),        # 1x1 channel tail (Cin=40) masked per lane
    (1, 6, 6, 24, 16, 3, 1, "leaky", "none", 3),        # generic 3x3 (Cin=24)
]
CASES += [(2, 12, 12, 64, 128, 3, 1, "leaky", "same", c) for c in range(NIG)]                  # every config
CASES += [(3, 10, 10, 128, 64, 1, 1, "elu", "none", c) for c in range(NIG)]                    # 1x1 on every config
CASES += [(2, 10, 10, 512, 256, 3, 1, "leaky", "same", c) for c in SPEC + (6,)]                # specialised / deep ring, long K
NE8, NE8N = NIG, 4   # 8-wave BM x 256 kernels (conv_8ph.hip): BM = 256, 224, 192, 160
for _c in range(NE8, NE8 + NE8N):
    CASES += [(2, 12, 12, 64, 128, 3, 1, "leaky", "same", _c),        # 9 K tiles, two m-tiles
              (3, 10, 10, 256, 320, 3, 1, "elu", "none", _c),         # two n-tiles, ragged M and N
              (1, 40, 40, 128, 256, 3, 1, "leaky", "same", _c),
              (2, 20, 20, 512, 128, 3, 1, "leaky", "up2", _c),
              (1, 3, 5, 64, 64, 3, 1, None, "none", _c),              # map smaller than a tile
              (1, 20, 20, 256, 512, 3, 2, "leaky", "none", _c),       # stride 2
              (2, 20, 20, 256, 128, 1, 1, "leaky", "none", _c),       # 1x1, 4 K tiles
              (1, 10, 10, 64, 512, 1, 1, "elu", "same", _c),          # 1x1, ONE K tile
              (1, 8, 8, 128, 64, 1, 1, None, "none", _c),             # 1x1, two K tiles
              (3, 9, 7, 192, 208, 3, 1, "leaky", "same", _c)]         # Cin = 3 x 64, odd map
# the streaming 3x3 kernels of the 32 <-> 64-channel layers (conv_stream3.hip; chosen by the library for these shapes)
CASES += [(2, 16, 32, 64, 32, 3, 1, "leaky", "none", -1),     # 64 -> 32 (backward-data of b.s1.0.b), 4 x 2 tiles per image
          (1, 8, 48, 32, 64, 3, 1, "elu", "none", -1),        # 32 -> 64 (b.s1.0.b forward), 64-byte LDS rows
          (2, 16, 64, 32, 64, 3, 2, None, "none", -1),        # 32 -> 64 stride 2 (b.down1 forward): two-buffer ring
          (3, 40, 64, 32, 64, 3, 2, "leaky", "none", -1),
          (5, 44, 80, 64, 32, 3, 1, None, "none", -1),        # 55 tiles per image, 275 tiles: several ring turns per workgroup
          (32, 160, 160, 32, 64, 3, 1, "leaky", "none", -1)]  # the training step's own shape (12 800 tiles)


def test_config_table_size(cuda):
    from object_detector_amd import _lib
    assert _lib.load().od_conv_num_tile_cfgs() == NIG + NE8N


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_matches_oracle(cuda, case):
    from object_detector_amd import ops
    B, H, W, Cin, Cout, k, stride, act, resm, cfg = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    res = None
    if resm == "same":
        res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
    elif resm == "up2":
        res = rng.normal(0, 1, (B, Ho // 2, Wo // 2, Cout)).astype(np.float16)
    alpha = 0.1 if act == "leaky" else 1.0
    xt = torch.from_numpy(x).to(cuda)
    rt = torch.from_numpy(res).to(cuda) if res is not None else None
    out = ops.conv2d(xt, w.astype(np.float32), scale, bias, stride=stride, act=act, alpha=alpha, res=rt,
                     res_mode=resm, tile_cfg=cfg)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    ref = _ref(x.astype(np.float32), w.astype(np.float32), scale, bias, stride, act, alpha,
               None if res is None else res.astype(np.float32), resm == "up2")
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


RDIRECT_CASES = [  # B, H, W, Cin, Cout, k, stride, act, residual
    (2, 16, 32, 64, 128, 3, 2, "leaky", False), (1, 8, 32, 64, 128, 3, 2, "elu", False), (3, 24, 96, 64, 128, 3, 2, None, False),
    (2, 26, 64, 64, 128, 3, 2, "leaky", False), (4, 160, 160, 64, 128, 3, 2, "leaky", False),
    # pointwise forms (stages 1-2 in training: forward without, backward-data with the in-place residual sum)
    (2, 12, 32, 64, 32, 1, 1, "leaky", False), (1, 7, 48, 32, 64, 1, 1, None, True), (3, 9, 16, 128, 64, 1, 1, "elu", False),
    (2, 10, 64, 64, 128, 1, 1, None, True), (8, 160, 160, 64, 32, 1, 1, "leaky", False), (8, 160, 160, 32, 64, 1, 1, None, True)]


@pytest.mark.parametrize("case", RDIRECT_CASES, ids=str)
def test_conv_weights_resident_kernels(cuda, case, monkeypatch):
    """od_conv_rdirect (conv_rdirect.hip): all weights in LDS, the pixel operand read straight from global memory -- what
    od_conv2d_fwd picks for `b.down2` (3x3 stride 2, 64 -> 128) and for the pointwise layers of stages 1-2 on large maps (the
    size threshold is lifted here so that small maps reach it too).  vs the table kernel the library would otherwise take
    (same inputs: equal to 1 f16 ulp + accumulation-order noise) and, on the small cases, vs the f64 oracle."""
    from object_detector_amd import ops
    monkeypatch.setenv("OD_CONV_RDIRECT_MIN_PIXELS", "0")
    B, H, W, Cin, Cout, k, stride, act, with_res = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    alpha = 0.1 if act == "leaky" else 1.0
    Ho, Wo = H // stride, W // stride
    res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16) if with_res else None
    xt = torch.from_numpy(x).to(cuda)
    rt = torch.from_numpy(res).to(cuda) if with_res else None
    kw = dict(stride=stride, act=act, alpha=alpha, res=rt, res_mode="same" if with_res else "none")
    out = ops.conv2d(xt, w.astype(np.float32), scale, bias, tile_cfg=-1, **kw)
    gen = ops.conv2d(xt, w.astype(np.float32), scale, bias, tile_cfg=2, **kw)
    torch.cuda.synchronize()
    got, g = out.cpu().numpy().astype(np.float64), gen.cpu().numpy().astype(np.float64)
    tolg = 1e-3 * max(1.0, np.abs(g).max()) + 2.0 ** -10 * np.abs(g)
    assert (np.abs(got - g) <= tolg).all(), f"vs table kernel: max {np.abs(got - g).max()}"
    if B * H * W <= 20000:  # the f64 oracle on the CPU
        ref = _ref(x.astype(np.float32), w.astype(np.float32), scale, bias, stride, act, alpha,
                   None if res is None else res.astype(np.float32))
        err = np.abs(got - ref)
        tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
        assert got.shape == ref.shape and (err <= tol).all(), f"max err {err.max()}"


PW_CASES = []
for _c in range(NE8, NE8 + NE8N):  # every BM of the 8-wave kernel: the second layer runs in the epilogue
    PW_CASES += [(2, 20, 20, 128, 256, 3, 1, "leaky", "same", "leaky", _c),   # stage-3 block: 3x3 + residual, then the next 1x1
                 (1, 40, 40, 128, 256, 3, 2, "leaky", "none", "leaky", _c),   # the stride-2 conv that opens the stage
                 (3, 9, 7, 192, 256, 3, 1, "elu", "same", "elu", _c),         # ragged last tile (189 pixels), ELU on both
                 (2, 10, 10, 512, 256, 1, 1, "leaky", "up2", None, _c)]       # 1x1 first layer, upsampled residual, linear second
PW_CASES += [(2, 20, 20, 128, 256, 3, 1, "leaky", "same", "leaky", -1),        # library's choice: fused or two launches
             (2, 20, 20, 128, 256, 3, 1, "leaky", "same", "leaky", 4),        # table kernel: always two launches
             (2, 12, 12, 64, 128, 3, 1, "leaky", "same", "leaky", NE8),       # Cout != 256: two launches
             (8, 40, 40, 128, 256, 3, 1, "leaky", "same", "leaky", -2)]       # throughput-mode choice at a multi-tile size


@pytest.mark.parametrize("case", PW_CASES, ids=str)
def test_conv_with_consuming_pointwise_layer(cuda, case):
    """od_conv_desc.w2: the 1x1 layer that consumes the launch's output, inside the 8-wave kernel's epilogue (or as a
    second launch where the selected kernel cannot).  `out` must be BIT-identical to the launch without w2; `out2` is
    checked against the f64 oracle applied to the DEVICE's f16 `out` (same rounding points as two separate layers):
    1 f16 ulp + 1e-3 of the output scale."""
    from object_detector_amd import ops
    B, H, W, Cin, Cout, k, stride, act, resm, act2, cfg = case
    Cout2 = Cout // 2
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    w2 = (rng.normal(0, 1, (Cout2, 1, 1, Cout)) * np.sqrt(2.0 / Cout)).astype(np.float16)
    scale2 = rng.uniform(0.5, 1.5, Cout2).astype(np.float32)
    bias2 = rng.normal(0, 0.1, Cout2).astype(np.float32)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    res = None
    if resm == "same":
        res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
    elif resm == "up2":
        res = rng.normal(0, 1, (B, Ho // 2, Wo // 2, Cout)).astype(np.float16)
    alpha = 0.1 if act == "leaky" else 1.0
    alpha2 = 0.1 if act2 == "leaky" else 1.0
    xt = torch.from_numpy(x).to(cuda)
    rt = torch.from_numpy(res).to(cuda) if res is not None else None
    plain = ops.conv2d(xt, w.astype(np.float32), scale, bias, stride=stride, act=act, alpha=alpha, res=rt, res_mode=resm,
                       tile_cfg=cfg)
    out, out2 = ops.conv2d(xt, w.astype(np.float32), scale, bias, stride=stride, act=act, alpha=alpha, res=rt,
                           res_mode=resm, tile_cfg=cfg,
                           next_pointwise=(w2.astype(np.float32), scale2, bias2, act2, alpha2))
    torch.cuda.synchronize()
    assert torch.equal(out, plain), "the first layer's output must not depend on the second layer being attached"
    y = out.cpu().numpy().astype(np.float32)
    ref2 = _ref(y, w2.astype(np.float32), scale2, bias2, 1, act2, alpha2)
    got2 = out2.cpu().numpy().astype(np.float64)
    assert got2.shape == ref2.shape == (B, Ho, Wo, Cout2)
    err = np.abs(got2 - ref2)
    tol = 1e-3 * max(1.0, np.abs(ref2).max()) + 2.0 ** -10 * np.abs(ref2)
    assert (err <= tol).all(), f"max err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, k, stride, act, res, cfg, splitk
    (1, 10, 10, 512, 1024, 3, 1, "leaky", "same", 5, 0),    # batch-1 stage-5 shape, library-chosen split
    (1, 10, 10, 512, 1024, 3, 1, "leaky", "same", 0, 6),
    (1, 20, 20, 512, 256, 1, 1, "elu", "up2", 7, 4),        # 1x1, split over channels
    (1, 20, 20, 256, 512, 3, 2, "leaky", "none", 2, 0),     # stride 2
    (1, 10, 10, 256, 208, 3, 1, None, "none", 3, 5),        # ragged Cout, generic-capable config
    (2, 6, 6, 40, 72, 3, 1, "leaky", "same", 3, 3),         # non-uniform taps (Cin = 40) + split
    (1, 10, 10, 512, 1024, 3, 1, "leaky", "same", NE8, 6),   # 8-wave kernel, 12 K tiles per split
    (2, 10, 10, 512, 304, 3, 1, "elu", "none", NE8 + 2, 5),      # BM = 192, uneven split (72 tiles / 5)
    (1, 20, 20, 512, 256, 1, 1, "elu", "up2", NE8 + 1, 4),       # 1x1, 2 K tiles per split
], ids=str)
def test_conv_split_k(cuda, case):
    """split-K path: per-split f32 slabs + finish kernel summing them in a fixed order (bit-reproducible)."""
    from object_detector_amd import ops
    B, H, W, Cin, Cout, k, stride, act, resm, cfg, sk = case
    rng = np.random.default_rng(11)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    res = None
    if resm == "same":
        res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
    elif resm == "up2":
        res = rng.normal(0, 1, (B, Ho // 2, Wo // 2, Cout)).astype(np.float16)
    alpha = 0.1 if act == "leaky" else 1.0
    rt = torch.from_numpy(res).to(cuda) if res is not None else None
    out = ops.conv2d(torch.from_numpy(x).to(cuda), w.astype(np.float32), scale, bias, stride=stride, act=act, alpha=alpha,
                     res=rt, res_mode=resm, tile_cfg=cfg, splitk=sk)
    out2 = ops.conv2d(torch.from_numpy(x).to(cuda), w.astype(np.float32), scale, bias, stride=stride, act=act,
                      alpha=alpha, res=rt, res_mode=resm, tile_cfg=cfg, splitk=sk)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)  # deterministic
    got = out.cpu().numpy().astype(np.float64)
    ref = _ref(x.astype(np.float32), w.astype(np.float32), scale, bias, stride, act, alpha,
               None if res is None else res.astype(np.float32), resm == "up2")
    err = np.abs(got - ref)
    tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
    assert (err <= tol).all(), err.max()


@pytest.mark.parametrize("case", [
    # B, [(H, W)], Cin, Cout, act, out_f32, cfg
    (8, [(40, 40), (20, 20), (10, 10)], 256, 256, "elu", False, -2),     # the prediction tower at 8 x 320^2 (8-wave kernel)
    (8, [(40, 40), (20, 20), (10, 10)], 256, 208, None, True, -2),       # the prediction conv: f32, Cout = 208
    (3, [(24, 40), (12, 20), (6, 10)], 128, 256, "leaky", False, NE8),   # ragged last tiles in every segment, BM = 256
    (3, [(24, 40), (12, 20), (6, 10)], 128, 256, "leaky", False, NE8 + 3),   # BM = 160
    (2, [(12, 12), (6, 6)], 64, 128, "elu", False, NE8 + 1),             # two segments, second smaller than a tile
    (1, [(40, 40), (20, 20), (10, 10)], 256, 256, "elu", False, -1),     # batch 1: the library falls back to one launch per level
    (2, [(12, 12), (6, 6), (3, 3)], 64, 128, "elu", False, 4),           # an explicit table config: always one launch per level
], ids=str)
def test_conv_grouped_over_pyramid_levels(cuda, case):
    """od_conv_desc.nseg: the same 3x3 layer over several maps in ONE launch (8-wave kernel, every m-tile inside one segment)
    -- or as one launch per segment when another kernel is selected.  Every segment vs the f64 oracle, and bit-identical to
    the ordinary launch of that segment alone on the same kernel."""
    from object_detector_amd import ops
    B, dims, Cin, Cout, act, out_f32, cfg = case
    rng = np.random.default_rng(hash(str(case)) & 0xFFFF)
    xs = [rng.normal(0, 1, (B, h, w, Cin)).astype(np.float16) for h, w in dims]
    w = (rng.normal(0, 1, (Cout, 3, 3, Cin)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float16)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    alpha = 0.1 if act == "leaky" else 1.0
    xts = [torch.from_numpy(x).to(cuda) for x in xs]
    outs = ops.conv2d_grouped(xts, w.astype(np.float32), scale, bias, act=act, alpha=alpha, out_f32=out_f32, tile_cfg=cfg)
    torch.cuda.synchronize()
    for x, xt, o in zip(xs, xts, outs):
        ref = _ref(x.astype(np.float32), w.astype(np.float32), scale, bias, 1, act, alpha)
        got = o.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref)
        tol = 1e-3 * max(1.0, np.abs(ref).max()) + (2.0 ** -10 * np.abs(ref) if not out_f32 else 0)
        assert got.shape == ref.shape and (err <= tol).all(), f"max err {err.max()}"
        if cfg >= NE8:  # the same kernel family alone on this map: same K order per output element -> the same bits
            single = ops.conv2d(xt, w.astype(np.float32), scale, bias, act=act, alpha=alpha, out_f32=out_f32, tile_cfg=cfg)
            assert torch.equal(single, o)


def test_conv_f32_strided_output(cuda):
    """prediction conv: Cout=208 (not a tile multiple), f32 logits written into a slice of pred[B,P,26]."""
    from object_detector_amd import ops
    rng = np.random.default_rng(7)
    B, H, W, Cin, Cout = 2, 10, 10, 256, 208
    P_total, C = 3000, 26
    off = 400  # rows
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, 3, 3, Cin)) * 0.02).astype(np.float16)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    pred = torch.full((B, P_total, C), 7.0, dtype=torch.float32, device=cuda)
    sl = pred.view(-1)[off * C:]
    ops.conv2d(torch.from_numpy(x).to(cuda), w.astype(np.float32), np.ones(Cout, np.float32), bias, out_f32=True,
               out=sl, out_batch_stride=P_total * C, out_pix_stride=Cout)
    torch.cuda.synchronize()
    got = pred.cpu().numpy()
    ref = _ref(x.astype(np.float32), w.astype(np.float32), np.ones(Cout), bias, 1, None, 0.0).reshape(B, H * W * 8, C)
    np.testing.assert_allclose(got[:, off:off + H * W * 8], ref, rtol=1e-4, atol=1e-4)
    assert (got[:, :off] == 7.0).all() and (got[:, off + H * W * 8:] == 7.0).all()  # nothing else touched


@pytest.mark.parametrize("shape", [(2, 32, 32), (1, 40, 72), (3, 8, 32)])
def test_conv_first_matches_oracle(cuda, shape):
    from object_detector_amd import ops
    B, H, W = shape
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    w = (rng.normal(0, 1, (32, 3, 3, 3)) * np.sqrt(2.0 / 27)).astype(np.float16)
    scale = (rng.uniform(0.5, 1.5, 32) / 255.0).astype(np.float32)
    bias = rng.normal(0, 0.1, 32).astype(np.float32)
    out = ops.conv_first(torch.from_numpy(x).to(cuda), w.astype(np.float32), scale, bias, "leaky", 0.1)
    torch.cuda.synchronize()
    ref = _ref(x.astype(np.float32), w.astype(np.float32), scale, bias, 1, "leaky", 0.1)
    got = out.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref)
    assert (err <= 1e-3 + 2.0 ** -10 * np.abs(ref)).all(), err.max()


def test_upsample2x_add(cuda):
    from object_detector_amd import ops
    rng = np.random.default_rng(5)
    a = rng.normal(0, 1, (2, 8, 12, 64)).astype(np.float16)
    u = rng.normal(0, 1, (2, 4, 6, 64)).astype(np.float16)
    out = ops.upsample2x_add(torch.from_numpy(a).to(cuda), torch.from_numpy(u).to(cuda)).cpu().numpy()
    ref = (a.astype(np.float32) + np.repeat(np.repeat(u, 2, 1), 2, 2).astype(np.float32)).astype(np.float16)
    assert (out == ref).all()


def test_conv_rejects_bad_args(cuda):
    from object_detector_amd import ops, _lib
    x = torch.zeros((1, 4, 4, 6), dtype=torch.float16, device=cuda)  # Cin % 8 != 0
    with pytest.raises(_lib.OdError):
        ops.conv2d(x, np.zeros((8, 3, 3, 6), np.float32), np.ones(8), np.zeros(8))


def test_integration_md_stub_runs_a_conv(cuda):
    """INTEGRATION.md §2, both code blocks executed VERBATIM (only the library path is made absolute): the stub a pytoolkit
    maintainer would add must drive one fused conv + BN + LeakyReLU correctly through the C ABI."""
    import pathlib
    import re
    from object_detector_amd import _lib
    from object_detector_amd.net import pack_conv_weight
    md = (pathlib.Path(__file__).resolve().parent.parent / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    assert len(blocks) >= 2
    ns = {}
    exec(blocks[0].replace('C.CDLL("libodhip.so")', f'C.CDLL({str(_lib.LIB_PATH)!r})'), ns)
    exec(blocks[1], ns)
    B, H, Wd, Cin, Cout, k, stride = 2, 16, 16, 64, 128, 3, 2
    rng = np.random.default_rng(11)
    x = rng.normal(0, 1, (B, H, Wd, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    bias = rng.normal(0, 0.1, Cout).astype(np.float32)
    cp, kp = ns["C"].c_int(), ns["C"].c_int()
    ns["check"](ns["lib"].od_conv_weight_dims(Cout, Cin, k, ns["C"].byref(cp), ns["C"].byref(kp)))
    wp = pack_conv_weight(w.astype(np.float32))
    assert wp.shape == (cp.value, kp.value)
    pad = lambda v: np.concatenate([v, np.zeros(cp.value - len(v), np.float32)])
    xd, wd = torch.from_numpy(x).to(cuda), torch.from_numpy(wp).to(cuda)
    sd, bd = torch.from_numpy(pad(scale)).to(cuda), torch.from_numpy(pad(bias)).to(cuda)
    out = torch.empty((B, H // stride, Wd // stride, Cout), dtype=torch.float16, device=cuda)
    ns["conv2d_bn_leaky"](xd.data_ptr(), wd.data_ptr(), sd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, Wd, Cin, Cout, k,
                          stride, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = _ref(x.astype(np.float32), w.astype(np.float32), scale, bias, stride, "leaky", 0.1)
    got = out.cpu().numpy().astype(np.float64)
    assert (np.abs(got - ref) <= 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)).all()


def test_conv_grouped_random_shapes(cuda):
    """40 random grouped launches (2-3 maps of unrelated sizes down to 1 x 1, any batch, Cin a multiple of 64, ragged Cout,
    every tile height of the 8-wave kernel, f16 / f32 output): every segment bit-identical to its own ordinary launch on
    the same kernel, and nothing written outside the outputs (guard bands)."""
    from object_detector_amd import _lib, ops
    from object_detector_amd.net import Context, _stream_ptr, pack_conv_weight, pad_vec
    import ctypes as C
    rng = np.random.default_rng(123)
    ctx = Context.get(cuda)
    for it in range(40):
        B = int(rng.integers(1, 7))
        nseg = int(rng.integers(2, 4))
        dims = [(int(rng.integers(1, 41)), int(rng.integers(1, 41))) for _ in range(nseg)]
        Cin = int(rng.choice([64, 128, 192, 256]))
        Cout = int(rng.integers(1, 41)) * 8
        cfg = NE8 + int(rng.integers(0, NE8N))
        out_f32 = bool(rng.integers(0, 2))
        act = [None, "leaky", "elu"][int(rng.integers(0, 3))]
        alpha = 0.1 if act == "leaky" else 1.0
        w = (rng.normal(0, 1, (Cout, 3, 3, Cin)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float16).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        bias = rng.normal(0, 0.1, Cout).astype(np.float32)
        xs = [torch.from_numpy(rng.normal(0, 1, (B, h, wd, Cin)).astype(np.float16)).to(cuda) for h, wd in dims]
        wp = torch.from_numpy(pack_conv_weight(w)).to(cuda)
        sc = torch.from_numpy(pad_vec(scale, wp.shape[0])).to(cuda)
        bi = torch.from_numpy(pad_vec(bias, wp.shape[0])).to(cuda)
        dt = torch.float32 if out_f32 else torch.float16
        GUARD = 256
        bufs = [torch.full((B * h * wd * Cout + 2 * GUARD,), 777.0, dtype=dt, device=cuda) for h, wd in dims]
        d = _lib.ConvDesc()
        d.w, d.scale, d.bias = wp.data_ptr(), sc.data_ptr(), bi.data_ptr()
        d.B, d.Cin, d.Cout, d.ksize, d.stride = B, Cin, Cout, 3, 1
        d.act, d.alpha = _lib.ACT_ENUM[act], alpha
        d.out_dtype = _lib.OD_DT_F32 if out_f32 else _lib.OD_DT_F16
        d.tile_cfg, d.nseg = cfg, nseg
        for i, (x, (h, wd), bf) in enumerate(zip(xs, dims, bufs)):
            d.seg_x[i], d.seg_out[i], d.seg_H[i], d.seg_W[i] = x.data_ptr(), bf.data_ptr() + GUARD * bf.element_size(), h, wd
        _lib.check(ctx.lib.od_conv2d_fwd(ctx.handle, C.byref(d), _stream_ptr()), "grouped")
        torch.cuda.synchronize()
        for x, (h, wd), bf in zip(xs, dims, bufs):
            assert (bf[:GUARD] == 777.0).all() and (bf[-GUARD:] == 777.0).all(), (it, "guard band overwritten")
            got = bf[GUARD:-GUARD].view(B, h, wd, Cout)
            single = ops.conv2d(x, w, scale, bias, act=act, alpha=alpha, out_f32=out_f32, tile_cfg=cfg)
            assert torch.equal(single, got), (it, B, dims, Cin, Cout, cfg, out_f32)
