#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the CPU oracle (seeded inputs -> expected outputs).

The reference ships no fixtures for this path (parity unpinned), so these vectors pin the ORACLE:
tests/test_oracle_golden.py (CPU) re-computes them with the oracle, tests/test_gpu_golden.py feeds the same inputs to
the HIP kernels.  Data only: inputs and expected outputs, no code.  Run from the repo root:  python scripts/make_golden.py
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import assign as oassign  # noqa: E402
from oracle import loss as oloss  # noqa: E402
from oracle import network as onet  # noqa: E402
from oracle import nms as onms  # noqa: E402
from oracle import postprocess as opp  # noqa: E402

OUT = ROOT / "tests" / "golden"


def conv_cases():
    rng = np.random.default_rng(100)
    cases = {}
    for tag, (B, H, W, Cin, Cout, k, s, act) in {
        "c3x3_s1": (1, 8, 8, 32, 64, 3, 1, "leaky"),
        "c3x3_s2": (1, 8, 8, 64, 128, 3, 2, "leaky"),
        "c1x1": (2, 6, 6, 128, 64, 1, 1, "elu"),
    }.items():
        x = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)
        w = (rng.normal(0, 1, (Cout, k, k, Cin)) * np.sqrt(2.0 / (k * k * Cin))).astype(np.float16)
        scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
        bias = rng.normal(0, 0.1, Cout).astype(np.float32)
        Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
        res = rng.normal(0, 1, (B, Ho, Wo, Cout)).astype(np.float16)
        y = onet.conv_nhwc_numpy(x.astype(np.float64), w.astype(np.float64), s)  # independent f64 einsum path
        y = y * scale + bias
        a = 0.1 if act == "leaky" else 1.0
        y = np.where(y > 0, y, y * a) if act == "leaky" else np.where(y > 0, y, a * np.expm1(np.minimum(y, 0)))
        y = (y + res.astype(np.float64)).astype(np.float16)
        cases[tag] = dict(x=x, w=w, scale=scale, bias=bias, res=res, y=y, stride=np.int32(s),
                          act=np.bytes_(act.encode()))
    return cases


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    for tag, d in conv_cases().items():
        np.savez_compressed(OUT / f"conv_{tag}.npz", **d)

    # priors: first/last rows of every level + checksum of the full table at 320 and 640
    for size in (320, 640):
        pr = opp.make_priors((size, size))
        np.savez_compressed(OUT / f"priors_{size}.npz", head=pr[:64], tail=pr[-64:], count=np.int64(len(pr)),
                            sum=np.float64(pr.astype(np.float64).sum()),
                            absdiffsum=np.float64(np.abs(np.diff(pr.astype(np.float64), axis=0)).sum()))

    # tiny network forward: 64x64 input, batch 1, weights from the seed (never stored)
    params = onet.init_weights(seed=2)
    x = onet.synthetic_images(1, 64, seed=0)
    pred16 = onet.Runner(params, storage="f16").forward(x)
    pred32 = onet.Runner(params, storage="f32").forward(x)
    np.savez_compressed(OUT / "net_64.npz", x=x, pred_f16storage=pred16, pred_f32=pred32,
                        w_checksum=np.float64(sum(float(np.abs(v).sum()) for v in params.values())))

    # post-process + NMS on a 64x64-input sized prior set with clustered boxes
    rng = np.random.default_rng(7)
    priors = opp.make_priors((64, 64))
    P = len(priors)
    pred = rng.normal(0, 1.5, (2, P, 26)).astype(np.float32)
    conf, boxes = opp.head_postprocess(pred, priors)
    keep = []
    for b in range(2):
        k, *_ = onms.detect_image(conf[b], boxes[b], K=256, conf_threshold=0.01, iou_threshold=0.45, max_det=100)
        keep.append(np.pad(k, (0, 100 - len(k)), constant_values=-1))
    np.savez_compressed(OUT / "post_nms_64.npz", pred=pred, priors=priors, conf=conf, boxes=boxes,
                        keep=np.stack(keep).astype(np.int32))

    # assignment + loss
    gt = np.array([[0.10, 0.12, 0.55, 0.70], [0.40, 0.30, 0.95, 0.90], [0.02, 0.05, 0.12, 0.20]], np.float32)
    gc = np.array([3, 7, 11], np.int32)
    pr = opp.make_priors((128, 128))
    y, assigned = oassign.encode_truth(gt, gc, pr, 20)
    pred = np.random.default_rng(9).normal(0, 1, y.shape).astype(np.float32)
    losses, grad = oloss.loss_and_grad(pred, y, 20)
    np.savez_compressed(OUT / "assign_loss_128.npz", gt_boxes=gt, gt_classes=gc, y=y, assigned=assigned, pred=pred,
                        losses=losses, grad=grad.astype(np.float32))
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
