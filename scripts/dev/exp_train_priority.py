"""Does a high-priority main stream (dz -> dx chain) beside a normal-priority weight-gradient stream shorten the step?"""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from object_detector_amd import weights as W  # noqa: E402
from object_detector_amd.trainer import Trainer  # noqa: E402

sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

dev = torch.device("cuda:0")
B, S = 32, 320
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(dev)
anns = bench.bench_annotations(B, S, rng)


def run(tag, main_stream, wprio):
    with torch.cuda.stream(main_stream) if main_stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        tr = Trainer(W.random_init(2), B, (S, S), device=dev, lr=1e-3, momentum=0.9, loss_scale=1024.0)
        if wprio is not None:
            tr.wstream = torch.cuda.Stream(device=dev, priority=wprio)
        for _ in range(4):
            tr.step(x, anns)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                tr.step(x, anns)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"{tag:70s} {sorted(ts)[1]:.3f} ms/step", flush=True)
    del tr
    torch.cuda.empty_cache()


run("default: main = default stream, wgrad stream by the queue probe", None, None)
run("main = default stream, wgrad stream priority 0 (fresh)", None, 0)
for mp, wp in ((-1, 0), (-1, -1), (0, -1)):
    run(f"main = fresh stream priority {mp}, wgrad = fresh stream priority {wp}", torch.cuda.Stream(device=dev, priority=mp), wp)
run("default again", None, None)
