"""precision="mixed": north_star's "outputs within 1e-3 on logits" of the reference's fp32 path, as an ABSOLUTE bound on every
logit (max |dlogit| <= 1e-3 x logit scale) at the BASELINE sizes -- the default f16-storage plan meets it only statistically
(rms 2.2e-4 x scale, max 1.5e-3 x scale over 14 M logits of a random-init network; 3.3e-4 x scale with trained weights).

Which storage points carry the error was measured on the CPU (scripts/dev/attribute_logit_error*.py): 51 % of
the variance rides on the residual stream of stages 3-5 (rounded after each of its 20 blocks), 44 % in the last five layers of
the neck / prediction module.  The mixed plan keeps that stream in f32 (od_wide_add) and feeds the last 3x3 layers an f16
(hi, lo) operand pair; every multiply is still an f16 MFMA.  oracle.network.MixedPlan restates the plan on the CPU."""
import json
import pathlib

import numpy as np
import pytest
import torch

from oracle import network as onet
from oracle import nms as onms
from oracle.compare import logit_stats

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _record(key, rec):
    f = ROOT / "gpurun_out" / "r03_mixed_precision.json"
    f.parent.mkdir(exist_ok=True)
    data = json.loads(f.read_text()) if f.exists() else {}
    data[key] = rec
    f.write_text(json.dumps(data, indent=1, sort_keys=True))


def test_wide_add_kernel(cuda):
    """od_wide_add: v = y (+ res); out32 = v, out16 = f16(v), [hi | lo] = [f16(v) | f16(v - f16(v))], bit-exact vs numpy."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context, _stream_ptr
    ctx = Context.get(cuda)
    rng = np.random.default_rng(0)
    for M, Cn, res_kind in [(1000, 256, "f32"), (77, 8, "f16"), (513, 1024, None), (3, 64, "f32")]:
        y = rng.normal(0, 3, (M, Cn)).astype(np.float32)
        res = None if res_kind is None else rng.normal(0, 3, (M, Cn)).astype(np.float32 if res_kind == "f32" else np.float16)
        v = y if res is None else (y + res.astype(np.float32)).astype(np.float32)
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        yt = torch.from_numpy(y).to(cuda)
        rt = None if res is None else torch.from_numpy(res).to(cuda)
        o32 = torch.zeros((M, Cn), dtype=torch.float32, device=cuda)
        o16 = torch.zeros((M, Cn), dtype=torch.float16, device=cuda)
        hl = torch.zeros((M, 2 * Cn), dtype=torch.float16, device=cuda)
        d = _lib.WideDesc()
        d.y, d.res = yt.data_ptr(), (rt.data_ptr() if rt is not None else None)
        d.out32, d.out16, d.out_hilo = o32.data_ptr(), o16.data_ptr(), hl.data_ptr()
        d.M, d.C, d.res_f32 = M, Cn, int(res_kind == "f32")
        _lib.check(ctx.lib.od_wide_add(ctx.handle, C.byref(d), _stream_ptr()), "od_wide_add")
        torch.cuda.synchronize()
        assert np.array_equal(o32.cpu().numpy(), v) and np.array_equal(o16.cpu().numpy(), hi)
        got = hl.cpu().numpy()
        assert np.array_equal(got[:, :Cn], hi) and np.array_equal(got[:, Cn:], lo)
        # the pair carries the value to ~2^-22 relative
        back = got[:, :Cn].astype(np.float32) + got[:, Cn:].astype(np.float32)
        assert np.abs(back - v).max() <= 2.0 ** -21 * np.abs(v).max()
        # in place on the stream (res == out32), outputs optional
        if res_kind == "f32":
            d.out32, d.out16, d.out_hilo = rt.data_ptr(), None, None
            _lib.check(ctx.lib.od_wide_add(ctx.handle, C.byref(d), _stream_ptr()), "od_wide_add")
            torch.cuda.synchronize()
            assert np.array_equal(rt.cpu().numpy(), v)
    d.out32 = None
    assert ctx.lib.od_wide_add(ctx.handle, C.byref(d), _stream_ptr()) != 0  # no output at all: rejected
    # FPN sum: every element adds its nearest-neighbour parent of the half-size f32 (or f16) map
    Bn, H, Wd, Cn = 3, 6, 10, 64
    for res_dt in (np.float32, np.float16):
        y = rng.normal(0, 3, (Bn, H, Wd, Cn)).astype(np.float32)
        half = rng.normal(0, 3, (Bn, H // 2, Wd // 2, Cn)).astype(res_dt)
        v = (y + np.repeat(np.repeat(half.astype(np.float32), 2, 1), 2, 2)).astype(np.float32)
        yt, ht = torch.from_numpy(y).to(cuda), torch.from_numpy(half).to(cuda)
        o32 = torch.zeros((Bn, H, Wd, Cn), dtype=torch.float32, device=cuda)
        hl = torch.zeros((Bn, H, Wd, 2 * Cn), dtype=torch.float16, device=cuda)
        d = _lib.WideDesc()
        d.y, d.res, d.out32, d.out_hilo = yt.data_ptr(), ht.data_ptr(), o32.data_ptr(), hl.data_ptr()
        d.M, d.C, d.res_f32, d.res_up2, d.H, d.W = Bn * H * Wd, Cn, int(res_dt is np.float32), 1, H, Wd
        _lib.check(ctx.lib.od_wide_add(ctx.handle, C.byref(d), _stream_ptr()), "od_wide_add(up2)")
        torch.cuda.synchronize()
        assert np.array_equal(o32.cpu().numpy(), v)
        assert np.array_equal(hl.cpu().numpy()[..., :Cn], v.astype(np.float16))
    d.H = 5
    assert ctx.lib.od_wide_add(ctx.handle, C.byref(d), _stream_ptr()) != 0  # odd map: rejected


@pytest.mark.parametrize("B,S", [(2, 96), (3, 160), (2, (96, 224))], ids=["2-96", "3-160", "2-96x224"])
def test_mixed_plan_small(cuda, B, S):
    from object_detector_amd.detector import ObjectDetector
    hw = (S, S) if isinstance(S, int) else S
    x = np.random.default_rng(0).integers(0, 256, size=(B,) + hw + (3,), dtype=np.uint8)
    od = ObjectDetector.synthetic(B, hw, seed=2, device=cuda, use_multi_gpu=False, precision="mixed", n_inflight=1)
    keep, cnt = od.predict_batch_device(torch.from_numpy(x).to(cuda), conf_threshold=0.01)
    torch.cuda.synchronize()
    got = od.net.pred.cpu().numpy()
    ref32 = onet.Runner(od.params, storage="f32").forward(x)
    refm = onet.MixedPlan(od.net.stream_stages, od.net.split, od.net.wide_fpn).runner(od.params).forward(x)
    rec = logit_stats(got, refm, ref32)
    print(json.dumps({k: rec[k] for k in ("logit_scale", "max_rel_scale", "rms_rel_scale", "max_dev_vs_fp32")}))
    assert rec["max_dev_vs_fp32"] <= 1e-3 * rec["logit_scale"], rec
    assert rec["rms_rel_scale"] <= 1.5e-4 and rec["rms_dev_vs_fp32"] <= 1.15 * rec["rms_f16oracle_vs_fp32"], rec
    assert any(n == "od_wide_add_k" for n in od.net.time_ops()[1])
    conf, boxes = od.post.conf.cpu().numpy(), od.post.boxes.cpu().numpy()
    for b in range(B):
        r, *_ = onms.detect_image(conf[b], boxes[b])
        assert int(cnt[b]) == len(r) and (keep[b, :len(r)].cpu().numpy() == r).all()


@pytest.mark.parametrize("B,S", [(32, 320), (16, 640)], ids=["32-320", "16-640"])
def test_mixed_plan_at_baseline_config(cuda, B, S):
    """BASELINE configs[1] / [2] in the mixed plan, 3 batches in flight (bench.py's default detector) AND one at a time:
    EVERY logit within 1e-3 x scale of the fp32 oracle; NMS indices bit-exact on the device's conf / boxes."""
    from object_detector_amd.detector import ObjectDetector
    x = onet.synthetic_images(B, S, seed=0)
    xt = torch.from_numpy(x).to(cuda)
    ref32 = None
    for inflight in (3, 1):
        od = ObjectDetector.synthetic(B, (S, S), seed=2, device=cuda, use_multi_gpu=False, precision="mixed", n_inflight=inflight)
        if ref32 is None:
            from conftest import oracle_logits
            ref32 = oracle_logits(B, S, "f32")
            refm = oracle_logits(B, S, ("mixed", tuple(od.net.stream_stages), tuple(od.net.split), od.net.wide_fpn))
        if inflight > 1:
            tickets = [od.submit(xt, conf_threshold=0.01) for _ in range(inflight)]
            outs = []
            for t in tickets:
                keep, cnt = od.collect(t)
                p = od._pipes[t]
                outs.append((p.net.pred.cpu().numpy(), keep.cpu().numpy(), cnt.cpu().numpy(), p.post.conf.cpu().numpy(),
                             p.post.boxes.cpu().numpy()))
            for o in outs[1:]:
                assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
            got, keep, cnt, conf, boxes = outs[0]
        else:
            keep, cnt = od.predict_batch_device(xt, conf_threshold=0.01)
            torch.cuda.synchronize()
            got, keep, cnt = od.net.pred.cpu().numpy(), keep.cpu().numpy(), cnt.cpu().numpy()
            conf, boxes = od.post.conf.cpu().numpy(), od.post.boxes.cpu().numpy()
        rec = logit_stats(got, refm, ref32)
        rec["max_dev_vs_fp32_rel_scale"] = rec["max_dev_vs_fp32"] / rec["logit_scale"]
        print(f"[{B}x{S} mixed, {inflight} in flight] max |dlogit| vs fp32 oracle {rec['max_dev_vs_fp32']:.3e} = "
              f"{rec['max_dev_vs_fp32_rel_scale']:.2e} x scale {rec['logit_scale']:.1f}; rms vs the CPU mixed plan "
              f"{rec['rms_rel_scale']:.2e} x scale; rms vs fp32: device {rec['rms_dev_vs_fp32']:.3e} / CPU mixed plan "
              f"{rec['rms_f16oracle_vs_fp32']:.3e}")
        _record(f"infer_{S}x{B}_inflight{inflight}", rec)
        assert rec["max_dev_vs_fp32"] <= 1e-3 * rec["logit_scale"], rec
        assert rec["rms_dev_vs_fp32"] <= 1.15 * rec["rms_f16oracle_vs_fp32"], rec
        assert rec["max_over_sigma"] <= rec["gaussian_max_over_sigma"] + 3.0, rec
        for b in range(B):
            r, *_ = onms.detect_image(conf[b], boxes[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
            assert cnt[b] == len(r) and (keep[b, :len(r)] == r).all(), f"image {b}: kept indices differ"
        del od
        torch.cuda.empty_cache()
