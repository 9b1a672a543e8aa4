#!/usr/bin/env python3
"""Experiment: one batch-B detector vs S concurrent batch-B/S detectors on S HIP streams (kernel tails / ramps of one
sub-batch overlap the main phases of the others).  usage: python scripts/dev/exp_streams.py [--batch 32] [--size 320]"""
import argparse
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd.detector import ObjectDetector  # noqa: E402


def bench(dets, xs, streams, steps, graph):
    def step():
        cur = torch.cuda.current_stream()
        for d, x, s in zip(dets, xs, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                d.predict_batch_device(x, 0.01, graph=graph)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--splits", default="1,2,4")
    ap.add_argument("--full", action="store_true", help="every stream runs the FULL batch (pipelining of whole steps)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for S in [int(v) for v in a.splits.split(",")]:
        b = a.batch if a.full else a.batch // S
        dets = [ObjectDetector.synthetic(b, (a.size, a.size), device="cuda:0", use_multi_gpu=False) for _ in range(S)]
        g = torch.Generator(device="cpu").manual_seed(0)
        xs = [torch.randint(0, 256, (b, a.size, a.size, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(S)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        for graph in (False, True):
            ms = bench(dets, xs, streams, a.steps, graph)
            print(f"splits={S} sub-batch={b} graph={graph}: {ms:.3f} ms/step  {b * S / ms * 1e3:.0f} img/s", flush=True)
        del dets


if __name__ == "__main__":
    main()
