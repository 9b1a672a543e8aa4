#!/usr/bin/env python3
"""Read the s_memtime stamps of the OD_CONV_DEBUG=32 build of od_conv_8ph (K tile 10, waves 0 and 4 of workgroup 0)."""
import ctypes as C
import os
import pathlib
import sys

os.environ["OD_CONV_DEBUG"] = "32"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import torch  # noqa: E402
import bench_conv  # noqa: E402
from object_detector_amd.net import Context  # noqa: E402

ctx = Context.get("cuda:0")
us, tf = bench_conv.run(ctx, 16, 64, 64, 256, 256, 3, 1, 37, reps=5, res=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 40)()
assert ctx.lib.od_debug_e8_stamps(buf) == 0
print(f"{us:.1f} us {tf:.1f} TF/s")
names = ["L0", "L1", "M0", "M1"]
t0 = min(buf[0], buf[20])
for g in range(2):
    st = list(buf[g * 20:(g + 1) * 20])
    print('  coarse: prologue %d  mainloop %d  epilogue %d cycles' % (st[17] - st[16], st[18] - st[17], st[19] - st[18]))
    print(f"wave row {g}:")
    for ph in range(4):
        print(f"  p{ph}: " + "  ".join(f"{n}={st[ph * 4 + k] - t0:6d}" for k, n in enumerate(names)))
