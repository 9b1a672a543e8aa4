#!/usr/bin/env python3
"""Host-only: JPEG decode + resize throughput of a thread pool vs a process pool (is the GIL or the box's CPU share the limit?)."""
import os, sys, time, tempfile, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from concurrent.futures import ThreadPoolExecutor, ProcessPoolExecutor
from PIL import Image

def load(p):
    img = Image.open(p).convert("RGB")
    return np.asarray(img.resize((320, 320), Image.BILINEAR), np.uint8).sum()

def main():
    d = tempfile.mkdtemp(prefix="od_jpg_")
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    paths = []
    for i in range(512):
        p = os.path.join(d, f"{i}.jpg")
        Image.fromarray(np.roll(base, i, 1)).resize((500, 375), Image.BILINEAR).save(p, quality=90)
        paths.append(p)
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    try:
        print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
    except Exception as e:
        print("cpu.max n/a", e)
    for n in (1, 4, 8, 16):
        with ThreadPoolExecutor(n) as ex:
            t0 = time.perf_counter(); list(ex.map(load, paths)); dt = time.perf_counter() - t0
        print(f"threads {n:2d}: {512 / dt:7.0f} img/s")
    for n in (4, 16):
        with ProcessPoolExecutor(n) as ex:
            list(ex.map(load, paths[:32]))
            t0 = time.perf_counter(); list(ex.map(load, paths, chunksize=8)); dt = time.perf_counter() - t0
        print(f"procs   {n:2d}: {512 / dt:7.0f} img/s")

if __name__ == "__main__":
    main()
