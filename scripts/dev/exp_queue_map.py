#!/usr/bin/env python3
"""Does the in-flight speedup survive other streams having been created first?  (HIP maps streams onto a few hardware queues.)"""
import sys, pathlib, time
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd.detector import ObjectDetector

ndummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dummies = [torch.cuda.Stream() for _ in range(ndummy)]
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(1, device="cuda:0")
od = ObjectDetector.synthetic(32, (320, 320), device="cuda:0", use_multi_gpu=False, n_inflight=3)
x = torch.randint(0, 256, (32, 320, 320, 3), dtype=torch.uint8, device="cuda:0")
for _ in range(6):
    od.submit(x, 0.01)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    od.submit(x, 0.01)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 60
print(f"dummy streams {ndummy}: {32 / dt:.0f} img/s  ({dt * 1e3:.3f} ms/step)")
