"""Throughput cost of precision="mixed" at the BASELINE sizes, for several (stream_stages, split) choices.
usage: python scripts/dev/exp_mixed_cost.py [B] [S]"""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from object_detector_amd import weights as W  # noqa: E402
from object_detector_amd.detector import ObjectDetector  # noqa: E402
from object_detector_amd.net import Net  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 320
dev = torch.device("cuda:0")
x = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device=dev)
params = W.random_init(2)


def bench(od, steps=60, warm=15):
    for _ in range(warm):
        od.submit(x)
    od.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        od.submit(x)
    od.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


plans = [("f16", None, None), ("mixed", (4, 5), ("n.lat4", "n.lat5", "n.out3", "n.out4", "h.t0", "h.out")),
         ("mixed", (3, 4, 5), ("n.out3", "n.out4", "h.t0", "h.out")), ("mixed", (3, 4, 5), ("h.t0", "h.out")),
         ("mixed", (3, 4, 5), ()), ("mixed", (4, 5), ("n.out3", "n.out4", "h.t0", "h.out")), ("mixed", (), ("n.out3", "n.out4", "h.t0", "h.out")),
         ("mixed", (3, 4, 5), ("n.lat3", "n.lat4", "n.lat5", "n.out3", "n.out4", "h.t0", "h.out"))]
for prec, st, sp in plans:
    import os
    if st is not None:
        os.environ["OD_MIXED_STREAM"] = ",".join(map(str, st))
        os.environ["OD_MIXED_SPLIT"] = ",".join(sp)
    for nin in (3, 1):
        od = ObjectDetector(params, B, (S, S), device=dev, use_multi_gpu=False, precision=prec, n_inflight=nin)
        ms = bench(od)
        print(f"{prec:6s} stream {st} split {sp}: {nin} in flight {ms:.3f} ms/batch = {B / ms * 1e3:.0f} img/s", flush=True)
        del od
        torch.cuda.empty_cache()
