#!/usr/bin/env python3
"""Micro-benchmark of od_bottleneck_fwd on the stage-1 / stage-2 shapes.  usage: bench_bneck.py [--batch 32] [--size 320]"""
import argparse
import ctypes as C
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import _lib  # noqa: E402
from object_detector_amd.net import Context, pack_conv_weight, pad_vec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", type=int, default=0)
    a = ap.parse_args()
    ctx = Context.get("cuda:0")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for ch, hw in ((64, a.size // 2), (128, a.size // 4)):
        if a.only and a.only != ch:
            continue
        x = torch.randn((a.batch, hw, hw, ch), device=dev).half()
        out = torch.empty_like(x)
        w1 = torch.from_numpy(pack_conv_weight(rng.normal(0, 0.1, (ch // 2, 1, 1, ch)).astype(np.float32))).to(dev)
        w3 = torch.from_numpy(pack_conv_weight(rng.normal(0, 0.05, (ch, 3, 3, ch // 2)).astype(np.float32))).to(dev)
        s1 = torch.ones(w1.shape[0], device=dev)
        b1 = torch.zeros(w1.shape[0], device=dev)
        s3 = torch.ones(w3.shape[0], device=dev)
        b3 = torch.zeros(w3.shape[0], device=dev)
        d = _lib.BneckDesc()
        d.x, d.out, d.w1, d.w3 = x.data_ptr(), out.data_ptr(), w1.data_ptr(), w3.data_ptr()
        d.scale1, d.bias1, d.scale3, d.bias3 = s1.data_ptr(), b1.data_ptr(), s3.data_ptr(), b3.data_ptr()
        d.B, d.H, d.W, d.C = a.batch, hw, hw, ch
        d.act, d.alpha = 1, 0.1
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            _lib.check(ctx.lib.od_bottleneck_fwd(ctx.handle, C.byref(d), C.c_void_p(s)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            _lib.check(ctx.lib.od_bottleneck_fwd(ctx.handle, C.byref(d), C.c_void_p(s)))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        m = a.batch * hw * hw
        fl = 2.0 * m * (ch * ch // 2 + 9 * ch * ch // 2)
        print(f"bneck C={ch:4d} M={m:8d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {2 * m * ch * 2 / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
