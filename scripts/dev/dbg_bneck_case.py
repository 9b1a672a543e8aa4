#!/usr/bin/env python3
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent / "tests"))
from object_detector_amd import ops
import test_gpu_bneck as T
case = eval(sys.argv[1])
B, H, W, C, act = case
rng = np.random.default_rng(hash(case) & 0xFFFF)
x = rng.normal(0, 1, (B, H, W, C)).astype(np.float16)
w1 = (rng.normal(0, 1, (C // 2, 1, 1, C)) * np.sqrt(2.0 / C)).astype(np.float16)
w3 = (rng.normal(0, 1, (C, 3, 3, C // 2)) * np.sqrt(2.0 / (9 * C // 2))).astype(np.float16)
s1 = rng.uniform(0.5, 1.5, C // 2).astype(np.float32); b1 = rng.normal(0, 0.1, C // 2).astype(np.float32)
s3 = rng.uniform(0.5, 1.5, C).astype(np.float32); b3 = rng.normal(0, 0.1, C).astype(np.float32)
alpha = 0.1 if act == "leaky" else 1.0
dev = torch.device("cuda:0")
out = ops.bottleneck(torch.from_numpy(x).to(dev), w1.astype(np.float32), s1, b1, w3.astype(np.float32), s3, b3, act=act, alpha=alpha)
got = out.cpu().numpy().astype(np.float64)
ref = T._ref(x.astype(np.float32), w1.astype(np.float32), s1, b1, w3.astype(np.float32), s3, b3, act, alpha)
err = np.abs(got - ref)
tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
bad = err > tol
print("bad", bad.sum(), "of", bad.size)
print("bad per channel:", bad.sum(axis=(0, 1, 2)).tolist())
print("bad per y:", bad.sum(axis=(0, 2, 3)).tolist())
print("bad per x:", bad.sum(axis=(0, 1, 3)).tolist())
