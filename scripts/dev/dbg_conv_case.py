#!/usr/bin/env python3
"""Debug helper: run one conv parity case and print where it differs from the oracle."""
import sys, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent / "tests"))
from object_detector_amd import ops
import test_gpu_conv as T

case = eval(sys.argv[1])
B, H, W, Cin, Cout, k, stride, act, resm, cfg = case
rng = np.random.default_rng(hash(case) & 0xFFFF)
x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
bias = rng.normal(0, 0.1, Cout).astype(np.float32)
Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
res = None
if resm == "same":
    res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
elif resm == "up2":
    res = rng.normal(0, 1, (B, Ho // 2, Wo // 2, Cout)).astype(np.float16)
alpha = 0.1 if act == "leaky" else 1.0
dev = torch.device("cuda:0")
rt = torch.from_numpy(res).to(dev) if res is not None else None
out = ops.conv2d(torch.from_numpy(x).to(dev), w.astype(np.float32), scale, bias, stride=stride, act=act, alpha=alpha, res=rt,
                 res_mode=resm, tile_cfg=cfg)
torch.cuda.synchronize()
got = out.cpu().numpy().astype(np.float64)
ref = T._ref(x.astype(np.float32), w.astype(np.float32), scale, bias, stride, act, alpha,
             None if res is None else res.astype(np.float32), resm == "up2")
err = np.abs(got - ref)
tol = 1e-3 * max(1.0, np.abs(ref).max()) + 2.0 ** -10 * np.abs(ref)
bad = np.argwhere(err > tol)
print("bad count", len(bad), "of", err.size)
g = got.reshape(-1, Cout); r = ref.reshape(-1, Cout)
badm = sorted(set(int(i) for i in np.argwhere((np.abs(g - r) > tol.reshape(-1, Cout)).any(1)).ravel()))
print("bad rows (m):", badm[:64])
badc = sorted(set(int(c) for c in bad[:, -1]))
print("bad channels:", badc)
for b_ in bad[:8]:
    print(tuple(b_), got[tuple(b_)], ref[tuple(b_)])
