#!/usr/bin/env python3
"""Is the training step host-bound?  Wall time the host needs to ENQUEUE one step (no synchronisation inside) vs the time
until the GPU has finished it, per phase.  usage: exp_train_host_time.py [size] [batch]"""
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
for _ in range(3):
    tr.step(x, anns)
torch.cuda.synchronize()
y, _n, _ = tr.pb.encode_batch(anns, return_device=True)
for name, fn in (("encode_batch", lambda: tr.pb.encode_batch(anns, return_device=True)), ("forward", lambda: tr.forward(x)),
                 ("loss", lambda: tr.loss(y)), ("backward", tr.backward), ("allreduce+sgd", lambda: (tr.allreduce(), tr.sgd())),
                 ("whole step", lambda: tr.step(x, anns))):
    hs, gs = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append(t1 - t0); gs.append(t2 - t0)
    print(f"{name:14s} host enqueue {np.median(hs) * 1e3:7.3f} ms   until GPU done {np.median(gs) * 1e3:7.3f} ms")
