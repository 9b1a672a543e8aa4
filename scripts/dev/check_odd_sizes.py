#!/usr/bin/env python3
"""End-to-end parity at input sizes and batch sizes no test and no bench line uses (shape-driven kernel selection: the
streaming / weights-resident / riding-1x1 kernels switch on and off with the map sizes): logits vs the f16-storage oracle with
the size-independent criteria of oracle/compare.py, kept indices bit-exact vs the oracle NMS fed the device's conf / boxes,
throughput plan (3 in flight) and latency plan, fused and layer-by-layer (OD_FUSE_BLOCKS=0 in a second process).
usage: check_odd_sizes.py"""
import os, pathlib, sys
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd.detector import ObjectDetector
from oracle import network as onet, nms as onms
from oracle.compare import assert_logits, logit_stats

cases = [(8, 352, 480), (3, 608, 608), (20, 256, 256), (5, 416, 416), (12, 512, 384)]
for B, H, W in cases:
    rng = np.random.default_rng(B * 1000 + H)
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    xt = torch.from_numpy(x).to("cuda:0")
    ref = ref32 = None
    for nin in (3, 1):
        od = ObjectDetector.synthetic(B, (H, W), seed=2, device="cuda:0", use_multi_gpu=False, n_inflight=nin)
        if ref is None:
            ref = onet.Runner(od.params, storage="f16").forward(x)
            ref32 = onet.Runner(od.params, storage="f32").forward(x)
        if nin == 3:
            t = od.submit(xt, conf_threshold=0.01)
            keep, cnt = od.collect(t)
            p = od._pipes[t]
            pred, conf, boxes = p.net.pred.cpu().numpy(), p.post.conf.cpu().numpy(), p.post.boxes.cpu().numpy()
            names = sorted(set(p.net.time_ops()[1]))
        else:
            keep, cnt = od.predict_batch_device(xt, conf_threshold=0.01)
            torch.cuda.synchronize()
            pred, conf, boxes = od.net.pred.cpu().numpy(), od.post.conf.cpu().numpy(), od.post.boxes.cpu().numpy()
            names = sorted(set(od.net.time_ops()[1]))
        rec = logit_stats(pred, ref, ref32)
        assert_logits(rec, f"{B}x{H}x{W} inflight {nin}")
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        for b in range(B):
            r, *_ = onms.detect_image(conf[b], boxes[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
            assert cnt[b] == len(r) and (keep[b, :len(r)] == r).all(), f"image {b}: kept indices differ"
        special = [n for n in names if any(k in n for k in ("rdirect", "stream3", "true>", "stem", "bneck"))]
        print(f"{B}x{H}x{W} fuse={os.environ.get('OD_FUSE_BLOCKS', '1')} inflight {nin}: rms {rec['rms_rel_scale']:.2e} max/sigma {rec['max_over_sigma']:.1f}  {special}", flush=True)
        del od
        torch.cuda.empty_cache()
print("odd sizes ok")
