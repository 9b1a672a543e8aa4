#!/usr/bin/env python3
"""Robustness sweep: whole predict step at several input sizes / batch sizes; fused plan vs the layer-by-layer plan
(OD_FUSE_BLOCKS=0) on the same weights, in-flight vs direct path.  Prints max |dlogit| and whether kept boxes agree."""
import os
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd.detector import ObjectDetector  # noqa: E402


def run(B, H, W, fuse):
    os.environ["OD_FUSE_BLOCKS"] = "1" if fuse else "0"
    od = ObjectDetector.synthetic(B, (H, W), seed=2, device="cuda:0", n_inflight=2)
    x = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to("cuda:0")
    keep, cnt = od.predict_batch_device(x, conf_threshold=0.01)
    torch.cuda.synchronize()
    pred = od.net.pred.float().cpu().numpy().copy()
    k1, c1 = keep.cpu().numpy().copy(), cnt.cpu().numpy().copy()
    t = od.submit(x, conf_threshold=0.01)
    k2, c2 = od.collect(t)
    same = bool((c2.cpu().numpy() == c1).all() and (k2.cpu().numpy() == k1).all())
    return pred, k1, c1, same


def main():
    cases = [(1, 320, 320), (5, 416, 416), (3, 320, 480), (2, 608, 608), (7, 224, 352), (16, 640, 640), (33, 320, 320), (1, 1024, 1024)]
    for B, H, W in cases:
        pf, kf, cf, sf = run(B, H, W, True)
        pu, ku, cu, su = run(B, H, W, False)
        ok = np.isfinite(pf).all() and np.isfinite(pu).all()
        d = float(np.abs(pf - pu).max())
        sc = float(np.abs(pu).max())
        print(f"B={B:3d} {H}x{W}: finite={ok} max|fused-unfused|={d:.3e} (scale {sc:.2f}) kept fused/unfused {int(cf.sum())}/{int(cu.sum())} "
              f"inflight==direct {sf and su}", flush=True)


if __name__ == "__main__":
    main()
