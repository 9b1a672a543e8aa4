#!/usr/bin/env python3
"""Overlap matrix of fresh streams (the probe of detector.stream_queue_sets) next to the measured step time of every
3-stream combination: does 'pairwise overlapping' predict the fastest set?"""
import itertools
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import detector as D  # noqa: E402

dev = torch.device("cuda:0")
ndummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
od = D.ObjectDetector.synthetic(32, (320, 320), device=dev, use_multi_gpu=False, n_inflight=3)
od._calibrated = True
cands = [p.stream for p in od._pipes] + [torch.cuda.Stream(device=dev) for _ in range(3)]
cur = torch.cuda.current_stream(dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(cands[0]):
    torch.cuda._sleep(1000)
    e0.record()
    torch.cuda._sleep(200000)
    e1.record()
e1.synchronize()
spin = int(2.0 * 200000 / e0.elapsed_time(e1))
print("spin cycles for 2 ms:", spin)
allst = cands + [cur]
print("overlap matrix (row = long kernel's stream, col = short kernel's stream; last = the caller's current stream):")
for a in allst:
    print("  ", " ".join("-" if a is b else ("1" if D._streams_overlap(a, b, spin) else "0") for b in allst))
x = torch.randint(0, 256, (32, 320, 320, 3), dtype=torch.uint8, device=dev)


def run(steps):
    for _ in range(steps):
        od.submit(x, 0.01)
    torch.cuda.synchronize()


for combo in itertools.combinations(range(len(cands)), 3):
    for p, ci in zip(od._pipes, combo):
        p.stream = cands[ci]
    run(6)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(30)
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    print(combo, f"{sorted(ts)[1]:.3f} ms/step")

for p, st in zip(od._pipes, cands[:3]):
    p.stream = st
D._QUEUE_SETS.clear()
od._calibrated = False
run(6)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    run(30)
    ts.append((time.perf_counter() - t0) / 30 * 1e3)
print("stream_queue_sets picked", [allst.index(p.stream) if p.stream in allst else "new" for p in od._pipes], f"{sorted(ts)[1]:.3f} ms/step")
