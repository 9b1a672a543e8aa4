#!/usr/bin/env python3
"""Random-shape stress of this round's new kernels against their generic twins (GPU box): the consuming 1x1 layer in the
8-wave kernel's epilogue vs two launches, od_tconv_64_32 vs the generic transposed path, the streaming first-layer weight
gradient vs torch.  usage: stress_new_kernels.py [cases per kernel]"""
import ctypes as C, pathlib, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import ops, train_ops as T, _lib
from object_detector_amd.net import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
rng = np.random.default_rng(12345)
NE8 = 43  # first 8-wave config index (tests/test_gpu_conv.py)
bad = 0
for it in range(n):
    B, H, W = int(rng.integers(1, 5)), int(rng.integers(3, 30)), int(rng.integers(3, 30))
    k, stride = (3, 1) if it % 3 else (1, 1)
    if it % 7 == 0: k, stride = 3, 2
    Cin = int(rng.choice([64, 128, 192]))
    cfg = NE8 + int(rng.integers(0, 4))
    x = torch.from_numpy(rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)).to(dev)
    w = (rng.normal(0, 1, (256, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16).astype(np.float32)
    sc, bi = rng.uniform(0.5, 1.5, 256).astype(np.float32), rng.normal(0, 0.1, 256).astype(np.float32)
    w2 = (rng.normal(0, 1, (128, 1, 1, 256)) * np.sqrt(2.0 / 256)).astype(np.float16).astype(np.float32)
    sc2, bi2 = rng.uniform(0.5, 1.5, 128).astype(np.float32), rng.normal(0, 0.1, 128).astype(np.float32)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    res = torch.from_numpy(rng.normal(0, 1, (B, Ho, Wo, 256)).astype(np.float16)).to(dev) if it % 2 else None
    y, t = ops.conv2d(x, w, sc, bi, stride=stride, act="leaky", alpha=0.1, res=res, res_mode="same" if res is not None else "none",
                      tile_cfg=cfg, next_pointwise=(w2, sc2, bi2, "leaky", 0.1))
    y0 = ops.conv2d(x, w, sc, bi, stride=stride, act="leaky", alpha=0.1, res=res, res_mode="same" if res is not None else "none",
                    tile_cfg=cfg)
    t0 = ops.conv2d(y0, w2, sc2, bi2, act="leaky", alpha=0.1, tile_cfg=3)
    torch.cuda.synchronize()
    ok = torch.equal(y, y0) and float((t.float() - t0.float()).abs().max()) <= 2e-3 * max(1.0, float(t0.float().abs().max()))
    if not ok:
        bad += 1
        print("POINTWISE MISMATCH", (B, H, W, Cin, k, stride, cfg), float((t.float() - t0.float()).abs().max()))
print(f"pointwise-in-epilogue: {n} random cases, {bad} mismatches")
bad2 = 0
for it in range(n):
    B, Hs, Ws = int(rng.integers(1, 5)), 4 * int(rng.integers(1, 12)), 16 * int(rng.integers(1, 6))
    dz = torch.from_numpy(rng.normal(0, 1, (B, Hs, Ws, 64)).astype(np.float16)).to(dev)
    wm = torch.from_numpy((rng.normal(0, 1, (64, 288)) * np.sqrt(2.0 / 288)).astype(np.float32)).to(dev)
    _wf, wb = T.pack_weights(wm, 64, 32, 3)
    ones, zeros = torch.ones(wb.shape[0], device=dev), torch.zeros(wb.shape[0], device=dev)
    a = T.conv_packed(dz, wb, ones, zeros, 64, 32, 3, stride=2, transposed=True)
    g = T.conv_packed(dz, wb, ones, zeros, 64, 32, 3, stride=2, transposed=True, tile_cfg=1)
    torch.cuda.synchronize()
    d = float((a.float() - g.float()).abs().max())
    if d > 1e-3 * max(1.0, float(g.float().abs().max())) + 2.0 ** -10 * float(g.float().abs().max()):
        bad2 += 1
        print("TCONV MISMATCH", (B, Hs, Ws), d)
print(f"od_tconv_64_32: {n} random cases, {bad2} mismatches")
bad3 = 0
ctx = Context.get(dev)
lib, h = ctx.lib, ctx.handle
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(n):
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(3, 40)), 32 * int(rng.integers(1, 5))
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    dz = rng.normal(0, 1, (B, H, W, 32)).astype(np.float16)
    xt = torch.tensor(x.astype(np.float64) / 255.0).permute(0, 3, 1, 2)
    wt = torch.zeros((32, 3, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(xt, wt, padding=1).backward(torch.tensor(dz.astype(np.float64)).permute(0, 3, 1, 2))
    ref = wt.grad.permute(0, 2, 3, 1).numpy().reshape(32, 27)
    nb = lib.od_conv_first_bwd_weight_workspace_bytes(h, B, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dw = torch.zeros((32, 27), dtype=torch.float32, device=dev)
    xd, dzd = torch.from_numpy(x).to(dev), torch.from_numpy(dz).to(dev)  # (named: a temporary would be freed before the launch)
    _lib.check(lib.od_conv_first_bwd_weight(h, xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), B, H, W, 32, 1.0 / 255.0,
                                            ws.data_ptr(), nb, s), "first wgrad")
    torch.cuda.synchronize()
    e = np.abs(dw.cpu().numpy() - ref).max()
    if e > 2e-3 * max(1.0, np.abs(ref).max()):
        bad3 += 1
        print("FIRST WGRAD MISMATCH", (B, H, W), e)
print(f"streaming first-layer weight gradient: {n} random cases, {bad3} mismatches")
sys.exit(1 if bad + bad2 + bad3 else 0)
