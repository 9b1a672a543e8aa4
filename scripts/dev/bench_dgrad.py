#!/usr/bin/env python3
"""Micro-benchmark of the backward-data convolutions of the 320x320 / batch-32 training step (od_conv2d_fwd on the
backward weight pack; stride-2 layers in transposed mode).  usage: bench_dgrad.py [--cfg N] [--reps 10]"""
import argparse
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import train_ops as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cfg", type=int, default=-1)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", type=int, default=-1, help="index of the one shape to run")
    ap.add_argument("--sweep", action="store_true", help="run every table config 0..29 on the selected shapes")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    # (Hout of the forward conv, Cin, Cout, k, stride): dz is [B, Hout, Hout, Cout], dx is [B, Hout*stride, .., Cin]
    shapes = [(160, 32, 64, 3, 2), (160, 32, 64, 3, 1), (80, 64, 128, 3, 2), (80, 64, 128, 3, 1), (40, 128, 256, 3, 2),
              (40, 128, 256, 3, 1), (20, 256, 512, 3, 2), (10, 512, 1024, 3, 2)]
    if a.only >= 0:
        shapes = [shapes[a.only]]
    cfgs = list(range(30)) if a.sweep else [a.cfg]
    for cfg in cfgs:
      a.cfg = cfg
      for Ho, Cin, Cout, k, stride in shapes:
       try:
        run_one(a, dev, Ho, Cin, Cout, k, stride)
       except RuntimeError as e:
        print(f"cfg {cfg}: {str(e)[:80]}")


def run_one(a, dev, Ho, Cin, Cout, k, stride):
    if True:
        dz = torch.randn((a.batch, Ho, Ho, Cout), device=dev).half()
        wm = torch.randn((Cout, k * k * Cin), device=dev) * 0.05
        _wf, wb = T.pack_weights(wm, Cout, Cin, k)
        ones = torch.ones(wb.shape[0], device=dev)
        zeros = torch.zeros(wb.shape[0], device=dev)
        out = T.conv_packed(dz, wb, ones, zeros, Cout, Cin, k, stride=stride, transposed=(stride == 2), tile_cfg=a.cfg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            T.conv_packed(dz, wb, ones, zeros, Cout, Cin, k, stride=stride, transposed=(stride == 2), out=out, tile_cfg=a.cfg)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        M = a.batch * Ho * Ho
        useful = 2.0 * M * Cin * Cout * k * k
        mb = (dz.numel() + out.numel()) * 2 / 1e6
        print(f"cfg {a.cfg:3d} dgrad Ho={Ho:4d} {Cout:4d}->{Cin:4d} k{k} s{stride}  {us:8.1f} us  {useful / us / 1e6:7.1f} TF/s (useful)  "
              f"{mb / us * 1e3:7.0f} GB/s (dz + dx)", flush=True)


if __name__ == "__main__":
    main()
