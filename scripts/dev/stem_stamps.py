#!/usr/bin/env python3
"""s_memtime stamps of od_stem (OD_CONV_DEBUG=64 build): 6th tile of workgroup 0, waves 0 and 5."""
import ctypes as C, os, pathlib, sys
os.environ["OD_CONV_DEBUG"] = "64"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from object_detector_amd import ops, weights as W
from object_detector_amd.net import Context
B, S = 32, 320
params = W.random_init(2)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, S, S, 3), dtype=np.uint8)).cuda()
s0, b0 = W.fold_bn(params, "b.conv0"); s0 = (s0 / np.float32(255)).astype(np.float32)
s3, b3 = W.fold_bn(params, "b.down1")
for _ in range(3):
    out = ops.stem(x, params["b.conv0.w"], s0, b0, params["b.down1.w"], s3, b3, act="leaky", alpha=0.1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = ops.stem(x, params["b.conv0.w"], s0, b0, params["b.down1.w"], s3, b3, act="leaky", alpha=0.1)
e1.record(); torch.cuda.synchronize()
print(f"stem {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per call (includes the wrapper's weight packing)")
ctx = Context.get("cuda:0")
buf = (C.c_ulonglong * 16)()
assert ctx.lib.od_debug_stem_stamps(buf) == 0
names = ["tile start", "producer done", "barrier1", "fetch issued", "consumer done", "epilogue done", "store_u done", "barrier2"]
for g in range(2):
    st = list(buf[g * 8:(g + 1) * 8])
    print("wave", 0 if g == 0 else 5, " ".join(f"{n}={st[k] - st[0]}" for k, n in enumerate(names)))
