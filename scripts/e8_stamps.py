#!/usr/bin/env python3
"""Read the s_memtime stamps of the OD_CONV_DEBUG=32 build of od_conv_8ph (K tile 10, waves 0 and 4 of workgroup 0)."""
import ctypes as C
import os
import pathlib
import sys

os.environ["OD_CONV_DEBUG"] = "32"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import torch  # noqa: E402
import bench_conv  # noqa: E402
from object_detector_amd.net import Context  # noqa: E402

ctx = Context.get("cuda:0")
us, tf = bench_conv.run(ctx, 16, 64, 64, 256, 256, 3, 1, 34, reps=5, res=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 40)()
assert ctx.lib.od_debug_e8_stamps(buf) == 0
print(f"{us:.1f} us {tf:.1f} TF/s")
names = ["load0", "issued", "dma_ok", "mfma0", "mfma1"]
for g in range(2):
    st = list(buf[g * 20:(g + 1) * 20])
    base = st[0]
    print(f"wave group {g}:")
    for ph in range(4):
        row = [st[ph * 5 + k] - base if st[ph * 5 + k] else -1 for k in range(5)]
        print(f"  p{ph}: " + "  ".join(f"{n}={v:6d}" for n, v in zip(names, row)))
print("group1.load0 - group0.load0 =", buf[20] - buf[0])
