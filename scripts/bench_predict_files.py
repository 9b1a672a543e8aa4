#!/usr/bin/env python3
"""End-to-end predict() on image FILES (JPEG decode + resize on the host, upload, network, NMS, results to host):
synthetic VOC-sized JPEGs.  usage: bench_predict_files.py [--n 512] [--threads 1,4,16]"""
import argparse
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--procs", default="", help="comma list: also run with OD_DECODE_PROCS=N")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--profile", action="store_true", help="cProfile the last run (main thread)")
    a = ap.parse_args()
    from PIL import Image
    from object_detector_amd.detector import ObjectDetector
    d = tempfile.mkdtemp(prefix="od_jpg_")
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    paths = []
    for i in range(a.n):
        img = Image.fromarray(np.roll(base, i, 1)).resize((500, 375), Image.BILINEAR)  # smooth: JPEG-typical content
        p = os.path.join(d, f"{i}.jpg")
        img.save(p, quality=90)
        paths.append(p)
    od = ObjectDetector.synthetic(32, (a.size, a.size), seed=2, device="cuda:0", use_multi_gpu=False)
    od.predict(paths[:64], conf_threshold=0.3)  # warm-up (stream calibration, first touch)
    runs = [("0", t) for t in a.threads.split(",") if t] + [(q, "1") for q in a.procs.split(",") if q]
    for q, t in runs:
        os.environ["OD_DECODE_PROCS"] = q
        os.environ["OD_DECODE_THREADS"] = t
        if q != "0":
            od.predict(paths[:64], conf_threshold=0.3)  # start the workers
        if a.profile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        r = od.predict(paths, conf_threshold=0.3)
        dt = time.perf_counter() - t0
        if a.profile:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
        print(f"decode {os.environ.get('OD_DECODE_PROCS', '0')} procs / threads {t:>3s}: {a.n / dt:8.1f} images/s end to end ({dt * 1e3 / a.n:.2f} ms per image, {len(r)} results)", flush=True)


if __name__ == "__main__":
    main()
