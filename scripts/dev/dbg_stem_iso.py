#!/usr/bin/env python3
"""od_stem / conv_first + down1 vs the oracle on identical inputs (32 x 320^2): where do they differ, by how much."""
import pathlib, sys
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import weights as W, ops
from oracle import network as onet

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 320
params = W.random_init(2)
x = onet.synthetic_images(B, S, 0)
xt = torch.from_numpy(x).cuda()
s0, b0 = W.fold_bn(params, "b.conv0"); s0 = (s0 / np.float32(255)).astype(np.float32)
s3, b3 = W.fold_bn(params, "b.down1")
w0, w3 = params["b.conv0.w"], params["b.down1.w"]
run = onet.Runner(params, storage="f32")
t_ref = run.first(x)
t16 = t_ref.astype(np.float16)
r = run.conv(t16.astype(np.float32), "b.down1", stride=2, act=("leaky", 0.1))
r16 = r.astype(np.float16).astype(np.float32)
t_dev = ops.conv_first(xt, w0, s0, b0, "leaky", 0.1)
td = t_dev.cpu().numpy()
print("conv_first vs oracle f16: mismatch frac", np.mean(td != t16), "max ulps", (np.abs(td.astype(np.float32) - t16.astype(np.float32)) / np.spacing(np.abs(t16)).astype(np.float32)).max())
two = ops.conv2d(t_dev, w3, s3, b3, stride=2, act="leaky", alpha=0.1).cpu().numpy().astype(np.float32)
one = ops.stem(xt, w0, s0, b0, w3, s3, b3, act="leaky", alpha=0.1).cpu().numpy().astype(np.float32)
rms = np.sqrt(np.mean(r.astype(np.float64) ** 2))
for nm, d in (("two-kernel", two), ("stem", one)):
    e = np.abs(d - r16)
    i = np.unravel_index(e.argmax(), e.shape)
    print(f"{nm}: mismatch frac {np.mean(d != r16):.4e}  max abs err {e.max():.3e} ({e.max() / rms:.2e} of rms {rms:.3f}) at {i}: dev {d[i]} ref {r[i]} ref16 {r16[i]}")
    big = e > 4 * np.spacing(np.abs(r16).astype(np.float16)).astype(np.float32)
    print("   elements off by > 4 own-ulps:", int(big.sum()), "of", e.size, " |ref| of those: median", np.median(np.abs(r[big])) if big.any() else None)
print("stem vs two-kernel: mismatch", np.mean(one != two), "max abs", np.abs(one - two).max())
