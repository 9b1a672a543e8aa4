#!/usr/bin/env python3
"""Timing-only ablation of the training step (two streams, as bench.py runs it): drop one group of C-ABI calls and see
what the step costs without it.  usage: exp_train_ablate.py [size] [batch]"""
import pathlib, sys, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from object_detector_amd import weights as W
from object_detector_amd.trainer import Trainer

size = int(sys.argv[1]) if len(sys.argv) > 1 else 320
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if size == 320 else 16)
dev = torch.device("cuda:0")
tr = Trainer(W.random_init(2), batch, (size, size), device=dev, lr=1e-3, momentum=0.9, loss_scale=1024.0)
rng = np.random.default_rng(1000)
x = torch.from_numpy(rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
anns = bench.bench_annotations(batch, size, rng)
real = tr.lib

class Proxy:
    def __init__(self, skip): self.skip = skip
    def __getattr__(self, name):
        fn = getattr(real, name)
        if name in self.skip:
            return lambda *a: 0
        return fn

def timed():
    for _ in range(3):
        tr.step(x, anns)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(x, anns)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10 * 1e3)
    return float(np.median(ts))

groups = {"nothing": (), "weight gradients": ("od_conv2d_bwd_weight_slabs", "od_wgrad_reduce_multi", "od_conv_first_bwd_weight"),
          "slab reduce only": ("od_wgrad_reduce_multi",),
          "BatchNorm backward": ("od_bn_bwd",), "BatchNorm forward (stats + scale/act)": ("od_bn_stats", "od_scale_act"),
          "sgd + pack": ("od_sgd_step_multi", "od_pack_weights_multi", "od_grad_nonfinite")}
base = None
for k, skip in groups.items():
    tr.lib = Proxy(set(skip))
    t = timed()
    if base is None: base = t
    print(f"without {k:40s} {t:7.3f} ms/step  ({base - t:+.3f})", flush=True)
