#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient kernel (slab form, as the trainer calls it) on the layer shapes of the
320x320 / batch-32 training step.  usage: bench_wgrad.py [--reps 10]"""
import argparse
import ctypes as C
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import _lib  # noqa: E402
from object_detector_amd.net import Context  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    ctx = Context.get("cuda:0")
    lib, h = ctx.lib, ctx.handle
    dev = torch.device("cuda:0")
    # (H of the input, Cin, Cout, k, stride)
    shapes = [(320, 32, 64, 3, 2), (160, 64, 32, 1, 1), (160, 32, 64, 3, 1), (160, 64, 128, 3, 2), (80, 128, 64, 1, 1),
              (80, 64, 128, 3, 1), (80, 128, 256, 3, 2), (40, 256, 128, 1, 1), (40, 128, 256, 3, 1), (40, 256, 512, 3, 2),
              (20, 512, 256, 1, 1), (20, 256, 512, 3, 1), (20, 512, 1024, 3, 2), (10, 1024, 512, 1, 1), (10, 512, 1024, 3, 1),
              (40, 256, 256, 3, 1), (20, 256, 256, 3, 1), (10, 256, 256, 3, 1), (40, 256, 208, 3, 1), (40, 256, 256, 1, 1),
              (20, 512, 256, 1, 1), (10, 1024, 256, 1, 1)]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = 0.0
    for H, Cin, Cout, k, stride in shapes:
        Ho = H // stride
        x = torch.randn((a.batch, H, H, Cin), device=dev).half()
        dz = torch.randn((a.batch, Ho, Ho, Cout), device=dev).half()
        sp = lib.od_conv2d_bwd_weight_splits(h, a.batch, H, H, Cin, Cout, k, stride)
        slabs = torch.empty(sp * Cout * k * k * Cin, dtype=torch.float32, device=dev)

        def run():
            _lib.check(lib.od_conv2d_bwd_weight_slabs(h, x.data_ptr(), dz.data_ptr(), slabs.data_ptr(), a.batch, H, H, Cin,
                                                      Cout, k, stride, s))
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        fl = 2.0 * a.batch * Ho * Ho * Cout * k * k * Cin
        tot += us
        print(f"wgrad H={H:4d} {Cin:4d}->{Cout:4d} k{k} s{stride} splits {sp:3d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  "
              f"slabs {slabs.numel() * 4 / 1e6:6.1f} MB", flush=True)
    print(f"sum {tot:.0f} us")


if __name__ == "__main__":
    main()
