#!/usr/bin/env python3
"""Micro-benchmark of od_conv2d_fwd tile configurations on the Darknet53 layer shapes (MI355X tuning tool).
usage: python scripts/dev/bench_conv.py [--batch 32] [--size 320] [--shapes big|net|all] [--cfgs 0,1,2,3]"""
import argparse
import ctypes as C
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import _lib  # noqa: E402
from object_detector_amd.net import Context, pack_conv_weight  # noqa: E402


def run(ctx, B, H, W, Cin, Cout, k, stride, cfg, reps=20, res=True, splitk=1):
    dev = torch.device("cuda:0")
    x = torch.randn((B, H, W, Cin), device=dev).half()
    w = np.random.default_rng(0).normal(0, 0.05, (Cout, k, k, Cin)).astype(np.float32)
    wp = torch.from_numpy(pack_conv_weight(w)).to(dev)
    sc = torch.ones(wp.shape[0], device=dev)
    bi = torch.zeros(wp.shape[0], device=dev)
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=dev)
    r = torch.randn((B, Ho, Wo, Cout), device=dev).half() if res else None
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.bias, d.out = x.data_ptr(), wp.data_ptr(), sc.data_ptr(), bi.data_ptr(), out.data_ptr()
    d.res = r.data_ptr() if res else None
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = B, H, W, Cin, Cout, k, stride
    d.act, d.alpha, d.res_mode, d.out_dtype, d.tile_cfg = 1, 0.1, 1 if res else 0, 0, cfg
    if splitk != 1:
        ws = torch.empty(32 * B * Ho * Wo * Cout, dtype=torch.float32, device=dev)
        d.splitk, d.splitk_workspace, d.splitk_workspace_bytes = splitk, ws.data_ptr(), ws.numel() * 4
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(ctx.lib.od_conv2d_fwd(ctx.handle, C.byref(d), C.c_void_p(s)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(ctx.lib.od_conv2d_fwd(ctx.handle, C.byref(d), C.c_void_p(s)))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flops = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    return us, flops / us / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--shapes", default="net")
    ap.add_argument("--cfgs", default="")
    ap.add_argument("--nores", action="store_true")
    ap.add_argument("--splitk", type=int, default=1)
    ap.add_argument("--custom", default="", help="H,Cin,Cout,k,stride: one extra shape (used with --shapes custom)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    ctx = Context.get("cuda:0")
    ncfg = ctx.lib.od_conv_num_tile_cfgs()
    cfgs = [int(c) for c in a.cfgs.split(",")] if a.cfgs else list(range(ncfg))
    S, B = a.size, a.batch
    net = [  # (name, H, Cin, Cout, k, stride)
        ("down1", S, 32, 64, 3, 2), ("s1.a", S // 2, 64, 32, 1, 1), ("s1.b", S // 2, 32, 64, 3, 1),
        ("down2", S // 2, 64, 128, 3, 2), ("s2.a", S // 4, 128, 64, 1, 1), ("s2.b", S // 4, 64, 128, 3, 1),
        ("down3", S // 4, 128, 256, 3, 2), ("s3.a", S // 8, 256, 128, 1, 1), ("s3.b", S // 8, 128, 256, 3, 1),
        ("down4", S // 8, 256, 512, 3, 2), ("s4.a", S // 16, 512, 256, 1, 1), ("s4.b", S // 16, 256, 512, 3, 1),
        ("down5", S // 16, 512, 1024, 3, 2), ("s5.a", S // 32, 1024, 512, 1, 1), ("s5.b", S // 32, 512, 1024, 3, 1),
        ("n.out3", S // 8, 256, 256, 3, 1),
    ]
    big = [("big3x3", 160, 128, 256, 3, 1), ("big1x1", 160, 256, 128, 1, 1)]
    mid = [("mid3x3", 80, 128, 256, 3, 1), ("mid3x3n", 80, 256, 256, 3, 1)]
    w40 = [("w40k2304", 40, 256, 256, 3, 1)]
    small = [("n.lat3", S // 8, 256, 256, 1, 1), ("n.lat4", S // 16, 512, 256, 1, 1), ("n.lat5", S // 32, 1024, 256, 1, 1),
             ("s3.a", S // 8, 256, 128, 1, 1), ("s4.a", S // 16, 512, 256, 1, 1), ("s5.a", S // 32, 1024, 512, 1, 1),
             ("h.t0.L1", S // 16, 256, 256, 3, 1), ("h.t0.L2", S // 32, 256, 256, 3, 1), ("h.out.L2", S // 32, 256, 208, 3, 1),
             ("s5.b", S // 32, 512, 1024, 3, 1), ("s4.b", S // 16, 256, 512, 3, 1)]
    head = [("h.out.L0", S // 8, 256, 208, 3, 1), ("h.t0.L0", S // 8, 256, 256, 3, 1), ("h.t0.L1", S // 16, 256, 256, 3, 1),
            ("h.out.L1", S // 16, 256, 208, 3, 1), ("n.lat3", S // 8, 256, 256, 1, 1)]
    custom = []
    if a.custom:
        h, ci, co, kk, st = (int(v) for v in a.custom.split(","))
        custom = [("custom", h, ci, co, kk, st)]
    shapes = {"custom": custom, "net": net, "big": big, "mid": mid, "w40": w40, "small": small, "head": head, "nethead": net + head, "all": net + big}[a.shapes]
    print(f"{'layer':8s} {'M':>8s} {'N':>5s} {'K':>5s} | " + " | ".join(f"cfg{c:<2d} us    TF/s" for c in cfgs))
    for name, H, Cin, Cout, k, st in shapes:
        row = []
        for c in cfgs:
            try:
                us, tf = run(ctx, B, H, H, Cin, Cout, k, st, c, res=(k == 3 and st == 1 and not a.nores), splitk=a.splitk,
                             reps=a.reps)
                row.append(f"{us:8.1f} {tf:7.1f}")
            except _lib.OdError:
                row.append(f"{'-':>8s} {'-':>7s}")
        Ho = H // st
        print(f"{name:8s} {B * Ho * Ho:8d} {Cout:5d} {k * k * Cin:5d} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
