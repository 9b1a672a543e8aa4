#!/usr/bin/env python3
"""Per-call table of one training step, single stream (OD_TRAIN_WSTREAM=0): every C-ABI call of forward / backward is
bracketed by events.  usage: OD_TRAIN_WSTREAM=0 train_layers.py [size] [batch]"""
import os, pathlib, sys
os.environ.setdefault("OD_TRAIN_WSTREAM", "0")
os.environ.setdefault("OD_TRAIN_BUCKET_MB", "0")
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from object_detector_amd import weights as W, _lib
from object_detector_amd.trainer import Trainer

size = int(sys.argv[1]) if len(sys.argv) > 1 else 320
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if size == 320 else 16)
dev = torch.device("cuda:0")
tr = Trainer(W.random_init(2), batch, (size, size), device=dev, lr=1e-3, momentum=0.9, loss_scale=1024.0)
rng = np.random.default_rng(1000)
x = torch.from_numpy(rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
anns = bench.bench_annotations(batch, size, rng)
for _ in range(3):
    tr.step(x, anns)
torch.cuda.synchronize()

recs = []
class Proxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("od_") or name.endswith("_bytes") or name.endswith("_splits") or name.endswith("_rows"):
            return fn
        def wrapped(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = fn(*a); e1.record()
            recs.append([name, None, e0, e1])
            return rc
        return wrapped
real_check = _lib.check
def check(rc, what=""):
    if recs and recs[-1][1] is None:
        recs[-1][1] = what
    return real_check(rc, what)
_lib.check = check
import object_detector_amd.trainer as T
tr.lib = Proxy(tr.lib)
reps = 3
acc = {}
order = []
for r in range(reps):
    recs.clear()
    tr.step(x, anns)
    torch.cuda.synchronize()
    for i, (name, what, e0, e1) in enumerate(recs):
        key = (i, name, what)
        if r == 0: order.append(key)
        acc.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
tot = {}
for key in order:
    i, name, what = key
    us = float(np.median(acc[key]))
    tot[name] = tot.get(name, 0.0) + us
    print(f"{i:4d} {name:32s} {str(what):34s} {us:8.1f}")
print("# totals per entry point (us per step):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"#   {k:34s} {v:9.1f}")
print(f"#   sum {sum(tot.values()):9.1f}")
