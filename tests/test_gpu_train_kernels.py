"""GPU parity of the training-side kernels (K11/K12) against torch-CPU autograd in fp32/fp64 on the same seeded inputs.
Tolerances: f16 storage of activations/gradients (rtol 2^-9) + f32 accumulation-order noise."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act_t(a, act, alpha):
    if act == "leaky":
        return F.leaky_relu(a, alpha)
    if act == "elu":
        return F.elu(a, alpha)
    return a


@pytest.mark.parametrize("shape", [(3, 6, 10), (4, 37, 29)], ids=["m180", "m4292"])  # one workgroup / several partial rows + ragged unroll tails
@pytest.mark.parametrize("C,act", [(32, "leaky"), (64, "leaky"), (256, "elu"), (1024, "leaky"), (208, None)])
def test_bn_forward_backward(cuda, C, act, shape):
    from object_detector_amd import train_ops as T
    rng = np.random.default_rng(C)
    B, H, W = shape
    alpha = 0.1 if act == "leaky" else 1.0
    z = rng.normal(0.3, 1.5, (B, H, W, C)).astype(np.float16)
    gamma = rng.uniform(0.5, 1.5, C).astype(np.float32)
    beta = rng.normal(0, 0.2, C).astype(np.float32)
    dy = rng.normal(0, 1, (B, H, W, C)).astype(np.float16)
    zt = torch.tensor(z.astype(np.float64), requires_grad=True)
    gt = torch.tensor(gamma.astype(np.float64), requires_grad=True)
    bt = torch.tensor(beta.astype(np.float64), requires_grad=True)
    mu = zt.mean((0, 1, 2)); var = zt.var((0, 1, 2), unbiased=False)
    xh = (zt - mu) / torch.sqrt(var + 1e-3)
    y = _act_t(gt * xh + bt, act, alpha)
    y.backward(torch.tensor(dy.astype(np.float64)))

    zd, dyd = torch.from_numpy(z).to(cuda), torch.from_numpy(dy).to(cuda)
    gd, bd = torch.from_numpy(gamma).to(cuda), torch.from_numpy(beta).to(cuda)
    mean, rstd, scale, shift = T.bn_stats(zd, gd, bd, 1e-3)
    np.testing.assert_allclose(mean.cpu().numpy(), mu.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rstd.cpu().numpy(), (1 / torch.sqrt(var + 1e-3)).detach().numpy(), rtol=1e-4)
    yd = T.scale_act(zd, scale, shift, act, alpha).cpu().numpy().astype(np.float64)
    yr = y.detach().numpy()
    assert (np.abs(yd - yr) <= 2e-3 + 2.0 ** -9 * np.abs(yr)).all()
    dz, dgamma, dbeta = T.bn_bwd(zd, dyd, scale, shift, mean, rstd, act, alpha, bn=True)
    np.testing.assert_allclose(dgamma.cpu().numpy(), gt.grad.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(dbeta.cpu().numpy(), bt.grad.numpy(), rtol=2e-3, atol=2e-3)
    dzr = zt.grad.numpy()
    dzd = dz.cpu().numpy().astype(np.float64)
    assert (np.abs(dzd - dzr) <= 3e-3 * max(1.0, np.abs(dzr).max()) + 2.0 ** -9 * np.abs(dzr)).all()


def test_bias_layer_backward(cuda):
    """prediction conv: no BN, linear: dz = dy, dbias = sum dy"""
    from object_detector_amd import train_ops as T
    rng = np.random.default_rng(1)
    z = rng.normal(0, 1, (2, 5, 5, 208)).astype(np.float16)
    dy = rng.normal(0, 1, (2, 5, 5, 208)).astype(np.float16)
    one = torch.ones(208, device=cuda); bias = torch.zeros(208, device=cuda)
    dz, dg, db = T.bn_bwd(torch.from_numpy(z).to(cuda), torch.from_numpy(dy).to(cuda), one, bias, None, None, None, 0.0,
                          bn=False)
    assert (dz.cpu().numpy() == dy).all()
    np.testing.assert_allclose(db.cpu().numpy(), dy.astype(np.float64).sum((0, 1, 2)), rtol=1e-4, atol=1e-3)


CONV_CASES = [  # B, H, W, Cin, Cout, k, stride
    (2, 12, 12, 64, 128, 3, 1),
    (2, 12, 12, 64, 128, 3, 2),
    (3, 10, 10, 128, 64, 1, 1),
    (2, 16, 16, 32, 64, 3, 1),     # Cin=32: column tile spans 4 taps in the weight gradient
    (2, 8, 8, 256, 208, 3, 1),     # ragged Cout (208)
    (1, 20, 20, 128, 256, 3, 2),   # stride 2: backward-data rows grouped by parity class, 100 rows per class (tiles straddle)
    (2, 32, 32, 64, 128, 3, 2),    # stride 2: whole tiles per parity class (taps skipped per tile)
    (1, 12, 28, 32, 64, 3, 2),     # stride 2, 32 gradient channels out, non-square
    (2, 6, 10, 1024, 512, 1, 1),
    # shapes that take the 8-wave 256 x 256 weight-gradient kernel (od_conv_wgrad_w8): whole tiles, ragged Ktot (6.75 tiles),
    # ragged Cout (208), stride 2, one tap spread over two half-tiles (Cin = 256)
    (2, 12, 12, 256, 256, 3, 1),
    (2, 8, 8, 192, 256, 3, 1),
    (3, 9, 9, 256, 208, 3, 1),
    (2, 12, 20, 128, 256, 3, 2),
    (2, 10, 10, 512, 256, 1, 1),
    (2, 16, 32, 32, 64, 3, 2),    # backward-data of the first stride-2 conv: the streaming kernel of conv_tconv.hip (one tile row)
    (3, 40, 64, 32, 64, 3, 2),    # 5 x 2 tiles per image, 30 tiles
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_backward(cuda, case):
    from object_detector_amd import train_ops as T
    B, H, W, Cin, Cout, k, stride = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    dz = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
    xt = torch.tensor(x.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w.astype(np.float64)).permute(0, 3, 1, 2).requires_grad_(True)
    y = F.conv2d(xt, wt, stride=stride, padding=k // 2)
    y.backward(torch.tensor(dz.astype(np.float64)).permute(0, 3, 1, 2))
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy()
    dw_ref = wt.grad.permute(0, 2, 3, 1).numpy().reshape(Cout, k * k * Cin)

    xd, dzd = torch.from_numpy(x).to(cuda), torch.from_numpy(dz).to(cuda)
    wm = torch.from_numpy(w.astype(np.float32).reshape(Cout, -1)).to(cuda)
    wf, wb = T.pack_weights(wm, Cout, Cin, k)
    # pack layout check (forward)
    assert (wf.cpu().numpy()[:Cout, :k * k * Cin] == w.reshape(Cout, -1)).all()
    # backward-data
    ones = torch.ones(wb.shape[0], device=cuda); zeros = torch.zeros(wb.shape[0], device=cuda)
    dx = T.conv_packed(dzd, wb, ones, zeros, Cout, Cin, k, stride=stride, transposed=(stride == 2))
    dxd = dx.cpu().numpy().astype(np.float64)
    assert dxd.shape == dx_ref.shape
    tol = 2e-3 * max(1.0, np.abs(dx_ref).max()) + 2.0 ** -9 * np.abs(dx_ref)
    assert (np.abs(dxd - dx_ref) <= tol).all(), np.abs(dxd - dx_ref).max()
    # the one-call export od_conv2d_bwd_data runs the same kernels: bit-identical
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    ctx = Context.get(cuda)
    dx1 = torch.empty_like(dx)
    _lib.check(ctx.lib.od_conv2d_bwd_data(ctx.handle, dzd.data_ptr(), wb.data_ptr(), None, dx1.data_ptr(), B, Ho, Wo, Cin,
                                          Cout, k, stride, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(dx1, dx)
    # gradient accumulation through the epilogue's residual input
    acc = torch.from_numpy(rng.normal(0, 1, dx_ref.shape).astype(np.float16)).to(cuda)
    dx2 = T.conv_packed(dzd, wb, ones, zeros, Cout, Cin, k, stride=stride, transposed=(stride == 2), res=acc,
                        res_mode="same").cpu().numpy().astype(np.float64)
    ref2 = dx_ref + acc.cpu().numpy().astype(np.float64)
    acc1 = acc.clone()  # in place: dx_accumulate aliases dx
    _lib.check(ctx.lib.od_conv2d_bwd_data(ctx.handle, dzd.data_ptr(), wb.data_ptr(), acc1.data_ptr(), acc1.data_ptr(), B, Ho,
                                          Wo, Cin, Cout, k, stride, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.array_equal(acc1.cpu().numpy().astype(np.float64), dx2)
    assert (np.abs(dx2 - ref2) <= tol + 2.0 ** -9 * np.abs(ref2) + 1e-3).all()
    # backward-weight (f32, atomics: order noise only)
    dw = T.conv_bwd_weight(xd, dzd, Cin, Cout, k, stride).cpu().numpy()[:Cout, :k * k * Cin]
    np.testing.assert_allclose(dw, dw_ref, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(dw_ref).max()))
    # accumulation into an existing buffer (shared prediction module)
    dw_t = T.conv_bwd_weight(xd, dzd, Cin, Cout, k, stride)
    T.conv_bwd_weight(xd, dzd, Cin, Cout, k, stride, dw=dw_t)
    np.testing.assert_allclose(dw_t.cpu().numpy()[:Cout, :k * k * Cin], 2 * dw_ref, rtol=2e-3,
                               atol=4e-3 * max(1.0, np.abs(dw_ref).max()))


def test_down2_and_sgd(cuda):
    from object_detector_amd import train_ops as T
    rng = np.random.default_rng(2)
    d = rng.normal(0, 1, (2, 8, 12, 64)).astype(np.float16)
    up = T.down2_sum_add(torch.from_numpy(d).to(cuda)).cpu().numpy()
    ref = d.astype(np.float32).reshape(2, 4, 2, 6, 2, 64).sum((2, 4))
    np.testing.assert_allclose(up.astype(np.float32), ref, rtol=2e-3, atol=2e-3)
    base = torch.from_numpy(rng.normal(0, 1, up.shape).astype(np.float16)).to(cuda)
    up2 = T.down2_sum_add(torch.from_numpy(d).to(cuda), base.clone()).cpu().numpy().astype(np.float32)
    np.testing.assert_allclose(up2, ref + base.cpu().numpy().astype(np.float32), rtol=4e-3, atol=4e-3)
    n = 10007
    w = rng.normal(0, 1, n).astype(np.float32); m = rng.normal(0, 1, n).astype(np.float32)
    g = rng.normal(0, 1, n).astype(np.float32)
    wd, md = torch.from_numpy(w).to(cuda), torch.from_numpy(m).to(cuda)
    T.sgd_step(wd, md, torch.from_numpy(g).to(cuda), lr=0.1, momentum=0.9, weight_decay=1e-4, inv_loss_scale=1 / 128)
    mr = 0.9 * m + (g / 128 + 1e-4 * w)
    np.testing.assert_allclose(md.cpu().numpy(), mr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(wd.cpu().numpy(), w - 0.1 * mr, rtol=1e-6, atol=1e-7)


def test_bn_fold_bit_exact_vs_numpy(cuda):
    """od_bn_fold = the loader's host-side fold (weights.fold_bn / oracle fold_bn), bit for bit."""
    import ctypes as C
    from object_detector_amd import _lib, weights as W
    from object_detector_amd.net import Context
    from oracle import network as onet
    ctx = Context.get(cuda)
    rng = np.random.default_rng(0)
    for Cn in (32, 208, 1024, 5):
        p = {"l.gamma": rng.uniform(0.1, 1.5, Cn).astype(np.float32), "l.beta": rng.normal(0, 0.1, Cn).astype(np.float32),
             "l.mean": rng.normal(0, 0.5, Cn).astype(np.float32), "l.var": rng.uniform(0.01, 3.0, Cn).astype(np.float32),
             "l.w": np.zeros((Cn, 1, 1, 8), np.float32)}
        d = {k: torch.from_numpy(v).to(cuda) for k, v in p.items()}
        sc, bi = torch.empty(Cn, device=cuda), torch.empty(Cn, device=cuda)
        _lib.check(ctx.lib.od_bn_fold(ctx.handle, d["l.gamma"].data_ptr(), d["l.beta"].data_ptr(), d["l.mean"].data_ptr(),
                                      d["l.var"].data_ptr(), W.BN_EPS, sc.data_ptr(), bi.data_ptr(), Cn,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        rs, rb = onet.fold_bn(p, "l")
        assert np.array_equal(sc.cpu().numpy(), rs) and np.array_equal(bi.cpu().numpy(), rb)
        rs2, rb2 = W.fold_bn(p, "l")
        assert np.array_equal(rs, rs2) and np.array_equal(rb, rb2)


@pytest.mark.parametrize("case", [(8, 40, 40, 128, 256, 3, 1), (4, 20, 20, 256, 512, 3, 1), (32, 10, 10, 512, 1024, 3, 1),
                                  (4, 40, 40, 256, 512, 3, 2), (16, 40, 40, 256, 128, 1, 1),
                                  # the thin kernel (32 -> 64 channels, rows of whole 32-pixel chunks): stride 1 and 2, one
                                  # chunk per workgroup up to several ring turns
                                  (2, 8, 32, 32, 64, 3, 1), (3, 12, 64, 32, 64, 3, 1), (2, 16, 64, 32, 64, 3, 2),
                                  (5, 40, 128, 32, 64, 3, 2), (16, 160, 160, 32, 64, 3, 1)], ids=str)
def test_weight_gradient_slab_path_many_splits(cuda, case):
    """The form the trainer uses: per-split f32 slabs (plain stores) + fixed-order reduce, at sizes where the pixel range is
    split over many workgroups (both kernels: the 256-wide one needs >= 60 chunks per workgroup, e.g. the 16 x 160 x 160 case).  vs torch conv2d weight gradient in
    f64; two runs are bit-identical (no atomics anywhere)."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    B, H, W, Cin, Cout, k, stride = case
    rng = np.random.default_rng(7)
    x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
    Ho, Wo = H // stride, W // stride
    dz = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
    xt = torch.tensor(x.astype(np.float64)).permute(0, 3, 1, 2)
    wt = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    F.conv2d(xt, wt, stride=stride, padding=k // 2).backward(torch.tensor(dz.astype(np.float64)).permute(0, 3, 1, 2))
    ref = wt.grad.permute(0, 2, 3, 1).numpy().reshape(Cout, k * k * Cin)
    ctx = Context.get(cuda)
    lib, h = ctx.lib, ctx.handle
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd, dzd = torch.from_numpy(x).to(cuda), torch.from_numpy(dz).to(cuda)
    sp = lib.od_conv2d_bwd_weight_splits(h, B, H, W, Cin, Cout, k, stride)
    assert sp >= 1
    count = Cout * k * k * Cin
    outs = []
    for _ in range(2):
        slabs = torch.full((sp * count,), float("nan"), dtype=torch.float32, device=cuda)  # every element must be written
        _lib.check(lib.od_conv2d_bwd_weight_slabs(h, xd.data_ptr(), dzd.data_ptr(), slabs.data_ptr(), B, H, W, Cin, Cout, k,
                                                  stride, s))
        e = _lib.WgradRed()
        e.dw_offset, e.count, e.slabs, e.nslabs = 0, count, slabs.data_ptr(), sp
        tbl = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).to(cuda)
        g = torch.empty(count, dtype=torch.float32, device=cuda)
        _lib.check(lib.od_wgrad_reduce_multi(h, tbl.data_ptr(), 1, g.data_ptr(), s))
        torch.cuda.synchronize()
        outs.append(g.cpu().numpy().reshape(Cout, -1))
    print(f"{case}: {sp} splits")
    assert np.array_equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0], ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128, 3, 1), (2, 16, 16, 32, 64, 3, 2), (3, 10, 10, 128, 64, 1, 1),
                                  (4, 40, 40, 128, 256, 3, 1), (2, 12, 12, 256, 208, 3, 1), (32, 10, 10, 512, 1024, 3, 1),
                                  (1, 6, 6, 24, 16, 3, 1), (2, 20, 20, 512, 256, 1, 1)], ids=str)
def test_bn_statistics_from_the_conv_epilogue(cuda, case):
    """od_conv_desc.bn_partials: the raw convolution's epilogue writes per-m-tile partial sums of the f16 values it stores, and
    od_bn_stats_from_partials turns them into the SAME mean / rstd / scale / shift as the separate od_bn_stats pass over z
    (sums of the same f16 values in another fixed order: equal to f32 rounding), z itself bit-identical to the plain launch;
    two runs are bit-identical."""
    import ctypes as C
    from object_detector_amd import _lib, weights as W
    from object_detector_amd.net import Context, pack_conv_weight
    B, H, Wd, Cin, Cout, k, stride = case
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.normal(0, 1, (B, H, Wd, Cin)).astype(np.float16)).to(cuda)
    w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16).astype(np.float32)
    wp = torch.from_numpy(pack_conv_weight(w)).to(cuda)
    ones, zeros = torch.ones(wp.shape[0], device=cuda), torch.zeros(wp.shape[0], device=cuda)
    ctx = Context.get(cuda)
    lib, h = ctx.lib, ctx.handle
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Ho, Wo = H // stride, Wd // stride
    M = B * Ho * Wo

    def desc(out, part):
        d = _lib.ConvDesc()
        d.x, d.w, d.scale, d.bias, d.out = x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), out.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = B, H, Wd, Cin, Cout, k, stride
        d.act, d.res_mode, d.out_dtype, d.tile_cfg, d.splitk = _lib.OD_ACT_LINEAR, _lib.OD_RES_NONE, _lib.OD_DT_F16, -1, 1
        if part is not None:
            d.bn_partials, d.bn_partials_bytes = part.data_ptr(), part.numel() * 4
        return d
    z0 = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=cuda)
    _lib.check(lib.od_conv2d_fwd(h, C.byref(desc(z0, None)), s))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32)).to(cuda)
    beta = torch.from_numpy(rng.normal(0, 0.1, Cout).astype(np.float32)).to(cuda)
    ref = [torch.empty(Cout, device=cuda) for _ in range(4)]
    ws = torch.empty(lib.od_bn_workspace_bytes(M, Cout), dtype=torch.uint8, device=cuda)
    if Cout % 8 == 0:
        _lib.check(lib.od_bn_stats(h, z0.data_ptr(), M, Cout, gamma.data_ptr(), beta.data_ptr(), W.BN_EPS, *(t.data_ptr() for t in ref),
                                   None, None, 0.99, ws.data_ptr(), ws.numel(), s))
    outs = []
    for _ in range(2):
        part = torch.full((((M + 63) // 64) * 2 * Cout,), float("nan"), dtype=torch.float32, device=cuda)
        z1 = torch.empty_like(z0)
        d = desc(z1, part)
        rows = lib.od_conv2d_fwd_bn_rows(h, C.byref(d))
        assert 0 < rows <= (M + 63) // 64
        _lib.check(lib.od_conv2d_fwd(h, C.byref(d), s))
        got = [torch.empty(Cout, device=cuda) for _ in range(4)]
        _lib.check(lib.od_bn_stats_from_partials(h, part.data_ptr(), rows, M, Cout, gamma.data_ptr(), beta.data_ptr(), W.BN_EPS,
                                                 *(t.data_ptr() for t in got), None, None, 0.99, s))
        torch.cuda.synchronize()
        assert torch.equal(z1, z0)
        assert torch.isfinite(part[:rows * 2 * Cout]).all()
        outs.append([t.cpu().numpy() for t in got])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    zf = z0.float().cpu().numpy().reshape(M, Cout).astype(np.float64)
    mu, var = zf.mean(0), zf.var(0)
    np.testing.assert_allclose(outs[0][0], mu, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(outs[0][1], 1 / np.sqrt(var + W.BN_EPS), rtol=2e-4)
    for a, r in zip(outs[0], ref):
        np.testing.assert_allclose(a, r.cpu().numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("count,nslabs", [(2048, 512), (2048, 3), (8192, 77), (73728, 20), (1179648, 3), (32768, 1),
                                          (216 * 8, 130)], ids=str)
def test_wgrad_slab_reduce_order_is_fixed_and_restated(cuda, count, nslabs):
    """od_wgrad_reduce_multi: the slab range is cut into G contiguous groups (G from count and nslabs only), each group
    is added in ascending slab order, the G partial sums in ascending group order -- restated in numpy f32, bit for bit."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    rng = np.random.default_rng(count + nslabs)
    slabs = (rng.normal(0, 1, (nslabs, count)) * 10.0 ** rng.integers(-3, 4, (nslabs, 1))).astype(np.float32)
    n4 = count // 4
    G = 1
    while G < 64 and n4 * G < 65536 and G * 2 <= nslabs:
        G *= 2
    per = -(-nslabs // G)
    ref = None
    for g in range(G):
        k0, k1 = g * per, min((g + 1) * per, nslabs)
        if k0 >= k1:
            break
        part = slabs[k0].copy()
        for k in range(k0 + 1, k1):
            part += slabs[k]
        ref = part if ref is None else ref + part
    ctx = Context.get(cuda)
    lib, h = ctx.lib, ctx.handle
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sd = torch.from_numpy(slabs).to(cuda)
    off = 16
    g = torch.full((count + 2 * off,), float("nan"), dtype=torch.float32, device=cuda)
    e = _lib.WgradRed()
    e.dw_offset, e.count, e.slabs, e.nslabs = off, count, sd.data_ptr(), nslabs
    tbl = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).to(cuda)
    _lib.check(lib.od_wgrad_reduce_multi(h, tbl.data_ptr(), 1, g.data_ptr(), s))
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    assert np.isnan(got[:off]).all() and np.isnan(got[off + count:]).all()
    assert np.array_equal(got[off:off + count], ref)


@pytest.mark.parametrize("case", [(2, 8, 16), (1, 4, 48), (5, 20, 32), (32, 160, 160)], ids=str)
def test_first_downsample_backward_data_streaming_kernel(cuda, case):
    """od_tconv_64_32 (conv_tconv.hip; what od_conv2d_fwd(transposed=1) runs for dZ 64 -> dX 32 channels) against the
    generic transposed path (an explicit tile_cfg keeps that one): same packed weights, same f16 inputs, a per-channel
    scale / bias through both epilogues -> equal up to the f32 accumulation order (1 f16 ulp + 2^-18 of the magnitude sum,
    bounded here by 1e-3 of the output scale); two runs of the new kernel are bit-identical.  The last case is the training
    step's own shape (32 x 160 x 160 -> 320 x 320: 12 800 tiles over 512 persistent workgroups, 25 ring turns each)."""
    from object_detector_amd import train_ops as T
    B, Hs, Ws = case
    rng = np.random.default_rng(B * 1000 + Hs)
    dz = torch.from_numpy(rng.normal(0, 1, (B, Hs, Ws, 64)).astype(np.float16)).to(cuda)
    w = (rng.normal(0, 1, (64, 3, 3, 32)) * np.sqrt(2.0 / 288)).astype(np.float32)
    wm = torch.from_numpy(w.reshape(64, -1)).to(cuda)
    _wf, wb = T.pack_weights(wm, 64, 32, 3)
    scale = torch.from_numpy(np.pad(rng.uniform(0.5, 1.5, 32).astype(np.float32), (0, wb.shape[0] - 32))).to(cuda)
    bias = torch.from_numpy(np.pad(rng.normal(0, 0.1, 32).astype(np.float32), (0, wb.shape[0] - 32))).to(cuda)
    a = T.conv_packed(dz, wb, scale, bias, 64, 32, 3, stride=2, transposed=True)
    b = T.conv_packed(dz, wb, scale, bias, 64, 32, 3, stride=2, transposed=True)
    g = T.conv_packed(dz, wb, scale, bias, 64, 32, 3, stride=2, transposed=True, tile_cfg=1)
    torch.cuda.synchronize()
    assert a.shape == (B, 2 * Hs, 2 * Ws, 32)
    assert torch.equal(a, b)
    af, gf = a.float(), g.float()
    tol = 1e-3 * max(1.0, float(gf.abs().max())) + 2.0 ** -10 * gf.abs()
    bad = (af - gf).abs() > tol
    assert not bool(bad.any()), f"{int(bad.sum())} elements differ, max {float((af - gf).abs().max())}"
    print(f"{case}: {float((a != g).float().mean()):.2e} of the elements differ in the last bit from the generic path")


@pytest.mark.parametrize("case", [(2, 8, 32), (3, 20, 64), (2, 12, 48), (1, 5, 96), (8, 96, 96)], ids=str)
def test_first_layer_weight_gradient(cuda, case):
    """od_conv_first_bwd_weight: dW[co][tap*3 + c] += in_scale * sum over pixels of dz[p][co] * x_u8[p shifted by tap][c],
    f32, on the streaming kernel (W % 32 == 0) and on the widened-copy path (W = 48) vs torch's conv2d weight gradient in
    f64; it ACCUMULATES into dw; two runs are bit-identical (fixed summation order, no atomics)."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    B, H, W = case
    rng = np.random.default_rng(H * 100 + W)
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    dz = rng.normal(0, 1, (B, H, W, 32)).astype(np.float16)
    xt = torch.tensor(x.astype(np.float64) / 255.0).permute(0, 3, 1, 2)
    wt = torch.zeros((32, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(xt, wt, padding=1).backward(torch.tensor(dz.astype(np.float64)).permute(0, 3, 1, 2))
    ref = wt.grad.permute(0, 2, 3, 1).numpy().reshape(32, 27)  # [co][(dy*3 + dx)*3 + c]
    ctx = Context.get(cuda)
    lib, h = ctx.lib, ctx.handle
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd, dzd = torch.from_numpy(x).to(cuda), torch.from_numpy(dz).to(cuda)
    nb = lib.od_conv_first_bwd_weight_workspace_bytes(h, B, H, W)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    base = rng.normal(0, 1, (32, 27)).astype(np.float32)
    outs = []
    for _ in range(2):
        dw = torch.from_numpy(base.copy()).to(cuda)
        ws.fill_(0xFF)  # NaN patterns: every slab element that is read must have been written
        _lib.check(lib.od_conv_first_bwd_weight(h, xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), B, H, W, 32, 1.0 / 255.0,
                                                ws.data_ptr(), nb, s), "od_conv_first_bwd_weight")
        torch.cuda.synchronize()
        outs.append(dw.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    got = outs[0].astype(np.float64) - base
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize("case", [(2, 8, 16), (1, 5, 48), (3, 20, 32), (32, 80, 80)], ids=str)
def test_second_downsample_backward_data_weights_resident_kernel(cuda, case, monkeypatch):
    """od_tconv_rdirect<128, 64> (conv_rdirect.hip; what od_conv2d_fwd(transposed=1) runs for dZ 128 -> dX 64 channels on large
    maps: `b.down2` backward-data) against the generic transposed path, with and without the in-place accumulation into an
    existing gradient: equal up to the f32 accumulation order; two runs are bit-identical."""
    from object_detector_amd import train_ops as T
    monkeypatch.setenv("OD_CONV_RDIRECT_MIN_PIXELS", "0")
    B, Hs, Ws = case
    rng = np.random.default_rng(B * 1000 + Hs + 7)
    dz = torch.from_numpy(rng.normal(0, 1, (B, Hs, Ws, 128)).astype(np.float16)).to(cuda)
    wm = torch.from_numpy((rng.normal(0, 1, (128, 576)) * np.sqrt(2.0 / 576)).astype(np.float32)).to(cuda)
    _wf, wb = T.pack_weights(wm, 128, 64, 3)
    ones, zeros = torch.ones(wb.shape[0], device=cuda), torch.zeros(wb.shape[0], device=cuda)
    acc = torch.from_numpy(rng.normal(0, 1, (B, 2 * Hs, 2 * Ws, 64)).astype(np.float16)).to(cuda)
    for res in (None, acc):
        kw = dict(stride=2, transposed=True, res=res, res_mode="same" if res is not None else "none")
        a = T.conv_packed(dz, wb, ones, zeros, 128, 64, 3, **kw)
        b = T.conv_packed(dz, wb, ones, zeros, 128, 64, 3, **kw)
        g = T.conv_packed(dz, wb, ones, zeros, 128, 64, 3, tile_cfg=1, **kw)
        torch.cuda.synchronize()
        assert a.shape == (B, 2 * Hs, 2 * Ws, 64) and torch.equal(a, b)
        af, gf = a.float(), g.float()
        tol = 1e-3 * max(1.0, float(gf.abs().max())) + 2.0 ** -10 * gf.abs()
        bad = (af - gf).abs() > tol
        assert not bool(bad.any()), f"{int(bad.sum())} elements differ, max {float((af - gf).abs().max())}"
