#!/usr/bin/env python3
"""What post-processing (decode + top-K + NMS, ~10 small launches) costs INSIDE the 3-in-flight pipeline: the bench step
with and without it (timing only).  usage: exp_post_cost.py [size] [batch]"""
import pathlib, sys
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from object_detector_amd import weights as W
from object_detector_amd.detector import ObjectDetector

size = int(sys.argv[1]) if len(sys.argv) > 1 else 320
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if size == 320 else 16)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
od = ObjectDetector(W.random_init(2), batch, (size, size), device=dev, n_inflight=3)
x = torch.from_numpy(np.random.default_rng(1000).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
step = lambda: od.submit(x, conf_threshold=0.01)
for tag in ("with post-processing", "without"):
    if tag == "without":
        for p in od._pipes:
            p.post.run = lambda *a, **k: None
    for rep in range(2):
        times = bench.timed_reps(step, 30, 5, 5, 1, dev)
        print(f"{tag:22s} {bench._median(times) / 30 * 1e3:.4f} ms/step", flush=True)
