"""Where do the 12.0 ms per step of scripts/train.py go when the bare training step is 10.7 ms?"""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))
import torch  # noqa: E402

import _common  # noqa: E402
import train as T  # noqa: E402
from object_detector_amd import od_gen, weights as W  # noqa: E402
from object_detector_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
B, S, N = 32, 320, 200
X, y = _common.shapes_dataset(256, seed=0)
tr = Trainer(T.init_for_training(W.random_init(2)), B, (S, S), device=dev, lr=0.01, lr_multipliers={"h.": 1 / 3})


def timed(name, batches):
    tr.fit(batches, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.fit(batches, N)
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t0) / N * 1e3:.2f} ms/step", flush=True)


gen = od_gen.create_generator((S, S), encode_truth=tr.pb.encode_truth_device, device=dev, on_device=True, device_cache=True)
g, _ = gen.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=0)
fixed = [next(g) for _ in range(8)]


def cycle():
    while True:
        yield from fixed


timed("8 pre-built device batches, cycled (no generator)", cycle())
timed("generator in the training thread, device cache", g)
g2, _ = gen.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=0, prefetch=2)
timed("generator thread, 2 ahead, device cache", g2)
g3, _ = gen.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=0, prefetch=4)
timed("generator thread, 4 ahead, device cache", g3)
# generator alone
t0 = time.perf_counter()
for _i, _b in zip(range(100), g):
    pass
torch.cuda.synchronize()
print(f"generator alone: {(time.perf_counter() - t0) / 100 * 1e3:.2f} ms/batch")
