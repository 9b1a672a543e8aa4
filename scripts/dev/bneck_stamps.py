#!/usr/bin/env python3
"""s_memtime stamps of od_bneck<64> (OD_CONV_DEBUG=32 build): 6th tile of workgroup 0, waves 0 and 5."""
import ctypes as C, os, pathlib, subprocess, sys
C_ = sys.argv[1] if len(sys.argv) > 1 else "64"
os.environ["OD_CONV_DEBUG"] = "32" if C_ == "64" else "33"
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import torch
sys.argv = [sys.argv[0], "--only", C_, "--reps", "3"]
import bench_bneck
bench_bneck.main()
from object_detector_amd.net import Context
ctx = Context.get("cuda:0")
buf = (C.c_ulonglong * 16)()
assert ctx.lib.od_debug_bneck_stamps(buf) == 0
names = ["tile start", "producer done", "barrier1", "dma issued", "consumer done", "epilogue done", "dma wait", "barrier2"]
for g in range(2):
    st = list(buf[g * 8:(g + 1) * 8])
    print("wave", 0 if g == 0 else 5, " ".join(f"{n}={st[k] - st[0]}" for k, n in enumerate(names)))
