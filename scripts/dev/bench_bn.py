#!/usr/bin/env python3
"""Micro-benchmark of the training-side elementwise / reduction kernels (od_bn_stats, od_scale_act, od_bn_bwd) on the
layer shapes of the 320x320 / batch-32 step; rotating buffer sets so that repeats do not sit in L2 / MALL.
usage: bench_bn.py [--reps 10]"""
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
    a = ap.parse_args()
    ctx = Context.get("cuda:0")
    lib, h = ctx.lib, ctx.handle
    dev = torch.device("cuda:0")
    shapes = [(3276800, 32), (819200, 64), (819200, 32), (204800, 128), (204800, 64), (51200, 256), (51200, 128),
              (12800, 512), (12800, 256), (3200, 1024), (3200, 512)]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, Cc in shapes:
        nset = max(2, min(8, int(600e6 // (M * Cc * 2 * 3))))
        z = [torch.randn((M, Cc), device=dev).half() for _ in range(nset)]
        dy = [torch.randn((M, Cc), device=dev).half() for _ in range(nset)]
        dz = [torch.empty((M, Cc), device=dev, dtype=torch.float16) for _ in range(nset)]
        f = lambda v: torch.full((Cc,), v, device=dev, dtype=torch.float32)
        gamma, beta, mean, rstd, scale, shift, rm, rv, dg, db = f(1), f(0), f(0), f(1), f(1), f(0), f(0), f(1), f(0), f(0)
        wsb = lib.od_bn_workspace_bytes(M, Cc) + 2 * Cc * 4
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

        def stats(i):
            _lib.check(lib.od_bn_stats(h, z[i].data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-3, mean.data_ptr(),
                                       rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                       0.99, ws.data_ptr(), wsb, s))

        def apply(i):
            _lib.check(lib.od_scale_act(h, z[i].data_ptr(), scale.data_ptr(), shift.data_ptr(), None, 0, dz[i].data_ptr(),
                                        1, 1, M, Cc, 1, 0.1, s))

        def bwd(i):
            _lib.check(lib.od_bn_bwd(h, z[i].data_ptr(), dy[i].data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                     mean.data_ptr(), rstd.data_ptr(), M, Cc, 1, 0.1, 1, dg.data_ptr(), db.data_ptr(),
                                     dz[i].data_ptr(), ws.data_ptr(), wsb, s))

        line = f"M={M:8d} C={Cc:5d} ({M * Cc * 2 / 1e6:6.1f} MB/tensor)"
        for name, fn, ntens in (("stats", stats, 1), ("scale_act", apply, 2), ("bn_bwd", bwd, 5)):
            for i in range(nset):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(a.reps):
                fn(r % nset)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            line += f"  {name} {us:7.1f} us {ntens * M * Cc * 2 / us / 1e3:6.0f} GB/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
