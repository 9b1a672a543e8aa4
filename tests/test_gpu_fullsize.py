"""GPU parity AT THE BASELINE.json CONFIG SIZES, on the exact plans bench.py times.

`pick_cfg` chooses different kernels at full size than at the small shapes of the other GPU tests (od_conv_8ph BM=224/256
for stages 3/4, od_stem, persistent od_bneck<64>, the throughput-mode tile choice `tile_cfg = -2` of plans that run beside
other batches in flight, the weight-gradient pixel splits), so every check here runs the device at the FULL batch:

  configs[1]  Darknet53 320x320 inference, batch 32          -> test_inference_at_baseline_config[32-320]
  configs[2]  Darknet53 640x640 inference, batch 16, NMS index bit-exact vs CPU -> test_inference_at_baseline_config[16-640]
  configs[3]  320x320 training step, 32 images per GPU        -> test_training_step_at_baseline_config[32-320]
  configs[4]  640x640 training step, 16 images per GPU        -> test_training_step_at_baseline_config[16-640]
  voc_validate.py:14-15,25-27 defaults (320x320, batch 16) through load_voc -> test_load_voc_round_trip_validate_defaults

The oracle runs the same images on the host cores (torch-CPU conv, seconds to a minute).  Numbers are also written to
gpurun_out/fullsize_parity.json (DESIGN.md §5 quotes them).
"""
import json
import pathlib
import time

import numpy as np
import pytest
import torch

from oracle import network as onet
from oracle import nms as onms
from oracle import postprocess as opp
from oracle.compare import assert_logits as _assert_logits, logit_stats as _logit_stats

pytestmark = pytest.mark.gpu

ROOT = pathlib.Path(__file__).resolve().parent.parent
FULL = [(32, 320), (16, 640)]


def _record(key, rec):
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    f = out / "fullsize_parity.json"
    data = json.loads(f.read_text()) if f.exists() else {}
    data[key] = rec
    f.write_text(json.dumps(data, indent=1, sort_keys=True))


def _bench_annotations(B, S, seed):
    """bench.py's VOC-shaped ground truth recipe."""
    from object_detector_amd.pb import ObjectsAnnotation
    rng = np.random.default_rng(seed)
    anns = []
    for _ in range(B):
        n = int(np.clip(1 + rng.poisson(1.5), 1, 10))
        c = rng.uniform(0, 1, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.05), np.log(0.9), (n, 2)))
        anns.append(ObjectsAnnotation(None, S, S, rng.integers(0, 20, n),
                                      np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)))
    return anns


def _f16_ulp(v):
    """f16 spacing at |v| (f32 array), subnormal spacing 2^-24 at the bottom."""
    return np.spacing(np.abs(v).astype(np.float16)).astype(np.float32)


def _abs_params(params, names):
    """Parameters whose folded epilogue is |scale| * conv(|x|, |w|) + |bias|: the magnitude sum of every term of an output."""
    q = {}
    for n in names:
        sc, bi = onet.fold_bn(params, n)
        q[n + ".w"] = np.abs(params[n + ".w"])
        q[n + ".gamma"], q[n + ".var"] = np.abs(sc), np.full_like(sc, 1.0 - onet.BN_EPS)  # -> scale = |sc| exactly enough
        q[n + ".beta"], q[n + ".mean"] = np.abs(bi), np.zeros_like(bi)
    return q


def _check_layers_in_isolation(net, params, tag, done=None):
    """Every op of the plan against the oracle ON THE DEVICE'S OWN INPUT of that op.  Given bit-identical inputs the only
    differences left are the f32 accumulation order and the last ulp of exp(), so the device's f16 output must equal the
    oracle's rounded value except where the pre-rounding value sits on a rounding boundary:
        |dev - f16(oracle)| <= 1 f16 ulp of the value + 2^-20 * (|scale| * sum |x||w| + |bias|)
    -- the second term (16 f32 epsilons of the magnitude sum) is the accumulation-order noise of a cancelling sum, which can
    exceed the own ulp of a near-zero output -- and only for a small fraction of the elements."""
    run = onet.Runner(params, storage="f32")
    names = [s_[0] for s_ in onet.layer_specs(20)]
    mag = onet.Runner(_abs_params(params, names), storage="f32", backbone_act=None, head_act=None)
    worst_frac, rows = 0.0, []
    for inf, kname in zip(net.op_info, net.time_ops()[1]):
        kind, name = inf["kind"], inf["name"]
        # `done`: (layer, kernel) pairs another plan of the same test already checked on its own inputs -- the second plan
        # only re-checks the layers it runs on a DIFFERENT kernel (the oracle conv of every op is what this test costs)
        if done is not None:
            if (name, kname) in done:
                continue
            done.add((name, kname))
        if kind == "conv_group":  # the shared prediction module over the three levels in one launch: segment by segment
            for si, (xt, out) in enumerate(zip(inf["xs"], inf["outs"])):
                xf = xt.cpu().numpy().astype(np.float32)
                r = run.conv(xf, name, act=inf["act"])
                A = mag.conv(np.abs(xf), name)
                if inf["out_f32"]:
                    off, rows_n = inf["pred_rows"][si]
                    d = net.pred[:, off:off + rows_n].cpu().numpy().reshape(r.shape)
                    err = float((np.abs(d - r) / (2.0 ** -20 * A)).max())
                    assert err <= 1.0, (name, si, err)
                    rows.append((f"{name}[{si}]", kname, "f32", err, 0.0))
                else:
                    d = out.cpu().numpy().astype(np.float32)
                    r16 = r.astype(np.float16).astype(np.float32)
                    tol = _f16_ulp(r16) + np.float32(2.0 ** -20) * A
                    mx = float((np.abs(d - r16) / tol).max())
                    frac = float(np.mean(d != r16))
                    rows.append((f"{name}[{si}]", kname, "f16", mx, frac))
                    assert mx <= 1.0 and frac <= 0.02, f"{tag} {name}[{si}] ({kname}): {mx:.2f} x tol, {frac:.3%} differ"
                    worst_frac = max(worst_frac, frac)
            continue
        x = inf["x"].cpu().numpy()
        if kind == "first":
            r = run.first(x)
            A = None
        elif kind == "stem":
            t = run.first(x).astype(np.float16).astype(np.float32)
            r = run.conv(t, "b.down1", stride=2, act=inf["act"])
            A = mag.conv(np.abs(t), "b.down1", stride=2)
        elif kind == "bneck":
            xf = x.astype(np.float32)
            t = run.conv(xf, name + ".a", act=inf["act"]).astype(np.float16).astype(np.float32)
            r = run.conv(t, name + ".b", act=inf["act"], res=xf)
            A = mag.conv(np.abs(t), name + ".b", res=np.abs(xf))
        else:
            xf = x.astype(np.float32)
            res = None if inf["res"] is None else inf["res"].cpu().numpy().astype(np.float32)
            r = run.conv(xf, name, stride=inf["stride"], act=inf["act"], res=res, res_up2=inf["res_mode"] == 2)
            A = mag.conv(np.abs(xf), name, stride=inf["stride"], res=None if res is None else np.abs(res),
                         res_up2=inf["res_mode"] == 2)
        if kind == "conv" and inf["out_f32"]:
            off, rows_n = inf["pred_rows"]
            d = net.pred[:, off:off + rows_n].cpu().numpy().reshape(r.shape)
            err = float((np.abs(d - r) / (2.0 ** -20 * A)).max())
            assert err <= 1.0, (name, err)
            rows.append((name, kname, "f32", err, 0.0))
            continue
        d = inf["out"].cpu().numpy().astype(np.float32)
        r16 = r.astype(np.float16).astype(np.float32)
        rms = float(np.sqrt(np.mean(r.astype(np.float64) ** 2)))
        tol = _f16_ulp(r16) + (np.float32(2.0 ** -20) * A if A is not None else np.float32(0))
        if kind in ("stem", "bneck"):
            # a fused op also rounds its hidden tensor to f16; where THAT rounding falls the other way (1e-4 of the hidden
            # elements, measured on od_conv_first) nine-plus outputs move by w * ulp(t) -- up to a few 1e-3 of the layer's
            # rms (scripts/dev/dbg_stem_iso.py: 787 of 13 M outputs, identical for the fused and the two-kernel path)
            tol = tol + np.float32(4e-3 * rms)
        err = np.abs(d - r16)
        frac = float(np.mean(d != r16))
        mx = float((err / tol).max())
        rows.append((name, kname, "f16", mx, frac))
        assert mx <= 1.0, f"{tag} {name} ({kname}): {float(err.max()):.3e} off the oracle on identical inputs (rms {rms:.3f}, {mx:.2f} x tol)"
        assert frac <= 0.02, f"{tag} {name} ({kname}): {frac:.3%} of the outputs differ from the oracle on identical inputs"
        worst_frac = max(worst_frac, frac)
        if inf.get("then"):  # the consuming 1x1 layer rode in the same op: its input is the op's own f16 output
            nx = inf["then"]
            r2 = run.conv(d, nx["name"], act=nx["act"])
            A2 = mag.conv(np.abs(d), nx["name"])
            d2 = nx["out"].cpu().numpy().astype(np.float32)
            r2_16 = r2.astype(np.float16).astype(np.float32)
            err2 = np.abs(d2 - r2_16)
            mx2 = float((err2 / (_f16_ulp(r2_16) + np.float32(2.0 ** -20) * A2)).max())
            frac2 = float(np.mean(d2 != r2_16))
            rows.append((nx["name"], kname + " (second layer)", "f16", mx2, frac2))
            assert mx2 <= 1.0, f"{tag} {nx['name']} (in the epilogue of {kname}): {float(err2.max()):.3e} off the oracle, {mx2:.2f} x tol"
            assert frac2 <= 0.02, f"{tag} {nx['name']} (in the epilogue of {kname}): {frac2:.3%} of the outputs differ"
            worst_frac = max(worst_frac, frac2)
    print(f"[{tag}] per-layer isolation: {len(rows)} ops, worst mismatch fraction {worst_frac:.2e}, "
          f"worst error / tolerance {max(r_[3] for r_ in rows):.2f}")
    for r_ in sorted(rows, key=lambda t: -t[4])[:5]:
        print("    ", r_)
    return rows


@pytest.mark.parametrize("B,S", FULL, ids=[f"{b}-{s}" for b, s in FULL])
def test_inference_at_baseline_config(cuda, B, S):
    """The default detector (3 batches in flight -> every pipeline's plan is built in throughput mode) and the
    one-at-a-time detector (latency-mode plan), full batch: logits vs the f16-storage oracle on ALL images, kept indices
    bit-exact vs the oracle NMS fed the device's conf / boxes for ALL images."""
    from object_detector_amd.detector import ObjectDetector
    x = onet.synthetic_images(B, S, seed=0)
    xt = torch.from_numpy(x).to(cuda)
    od = ObjectDetector.synthetic(B, (S, S), seed=2, device=cuda, use_multi_gpu=False)
    assert od.n_inflight == 3, "bench.py's default"
    from conftest import oracle_logits  # (od.params are W.random_init(2) = onet.init_weights(2): asserted in test_host_logic)
    t0 = time.perf_counter()
    ref = oracle_logits(B, S, "f16")
    t_oracle = time.perf_counter() - t0
    ref32 = oracle_logits(B, S, "f32")
    tickets = [od.submit(xt, conf_threshold=0.01) for _ in range(3)]  # the step bench.py times, once per pipeline
    outs = []
    for t in tickets:
        keep, cnt = od.collect(t)
        p = od._pipes[t]
        outs.append(dict(pred=p.net.pred.cpu().numpy(), conf=p.post.conf.cpu().numpy(), boxes=p.post.boxes.cpu().numpy(),
                         keep=keep.cpu().numpy(), cnt=cnt.cpu().numpy()))
    _ms, names = od.net.time_ops()
    kernels = sorted(set(names))
    print("kernels of the throughput-mode plan:", kernels)
    if S == 320:
        assert any(k.startswith("od_conv_8ph") for k in kernels) and any("od_stem" in k for k in kernels)
        assert any(k.startswith("od_conv_8ph") and k.endswith("true, false>") for k in kernels), "stage 3: 1x1 in the producer's epilogue"
        assert any(k.startswith("od_conv_8ph") and k.endswith("false, true>") for k in kernels), "the prediction module as one grouped launch per layer"
    for o in outs[1:]:  # the three pipelines share the weights and must agree bit for bit
        assert np.array_equal(o["pred"], outs[0]["pred"]) and np.array_equal(o["keep"], outs[0]["keep"])
    o = outs[0]
    rec = _logit_stats(o["pred"], ref, ref32)
    _assert_logits(rec, f"{B}x{S} throughput plan")
    checked = set()
    layer_rows = _check_layers_in_isolation(od._pipes[0].net, od.params, f"{B}x{S} throughput plan", checked)
    for b in range(B):  # configs[2]: "NMS index bit-exact vs CPU", every image
        r, *_ = onms.detect_image(o["conf"][b], o["boxes"][b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
        assert o["cnt"][b] == len(r) and (o["keep"][b, :len(r)] == r).all(), f"image {b}: kept indices differ"
    # the whole oracle end to end (its own logits -> conf -> NMS): reported; equality is not claimed where candidates are
    # closer than the logit tolerance
    pri = opp.make_priors((S, S))
    conf_o, boxes_o = opp.head_postprocess(ref, pri)
    same = 0
    for b in range(B):
        r, *_ = onms.detect_image(conf_o[b], boxes_o[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
        same += int(o["cnt"][b] == len(r) and (o["keep"][b, :len(r)] == r).all())
    print(f"[{B}x{S}] images whose kept list equals the all-oracle pipeline's: {same}/{B}")
    del od
    torch.cuda.empty_cache()
    # latency-mode plan (--inflight 1, tile_cfg = -1): different tile choices for stage 4 etc.
    od1 = ObjectDetector.synthetic(B, (S, S), seed=2, device=cuda, use_multi_gpu=False, n_inflight=1)
    keep1, cnt1 = od1.predict_batch_device(xt, conf_threshold=0.01)
    torch.cuda.synchronize()
    pred1 = od1.net.pred.cpu().numpy()
    kernels1 = sorted(set(od1.net.time_ops()[1]))
    print("kernels of the latency-mode plan:", kernels1)
    rec1 = _logit_stats(pred1, ref, ref32)
    _assert_logits(rec1, f"{B}x{S} latency plan")
    n_before = len(checked)
    _check_layers_in_isolation(od1.net, od1.params, f"{B}x{S} latency plan", checked)
    print(f"latency plan: {len(checked) - n_before} (layer, kernel) pairs not already covered by the throughput plan")
    conf1, boxes1 = od1.post.conf.cpu().numpy(), od1.post.boxes.cpu().numpy()
    keep1, cnt1 = keep1.cpu().numpy(), cnt1.cpu().numpy()
    for b in range(B):
        r, *_ = onms.detect_image(conf1[b], boxes1[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
        assert cnt1[b] == len(r) and (keep1[b, :len(r)] == r).all(), f"image {b}: kept indices differ (latency plan)"
    _record(f"infer_{S}x{B}", dict(throughput_plan=rec, latency_plan=rec1, images=B, nms_bit_exact_images=B,
                                   end_to_end_equal_images=same, oracle_seconds=round(t_oracle, 1), kernels=kernels,
                                   kernels_latency_plan=kernels1,
                                   layers_worst_mismatch=[list(map(str, r_)) for r_ in
                                                          sorted(layer_rows, key=lambda t: -t[4])[:5]]))


def _device_slope_masks(tr):
    """Which side of the LeakyReLU kink the DEVICE put every unit on: a = z * scale + shift > 0 (od_bn_bwd / od_scale_act,
    train.hip act_grad / act_fwd; separate mul and add, as that TU is compiled with -ffp-contract=off)."""
    masks = {}
    for n in tr.nodes:
        if n.act and n.act[0] == "leaky":
            a = n.z.float() * n.scale + n.shift
            masks[n.name] = (a > 0).cpu().numpy()
            del a
    return masks


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("B,S", FULL, ids=[f"{b}-{s}" for b, s in FULL])
def test_training_step_at_baseline_config(cuda, B, S):
    """One full-batch training step (the per-GPU shard of configs[3] / configs[4]) with bench.py's loss scale: logits, the
    three losses and the gradient of EVERY parameter segment vs the torch-CPU oracle on the same batch.  The oracle runs in
    f32 here (f64 autograd of 32 x 320^2 takes minutes; tests/test_oracle_golden.py pins f32 against f64 on a small
    case) and differentiates the device's LeakyReLU sign pattern (oracle/train_ref.py `slope_masks`); the units that sit
    on the other side of the kink than in the oracle's own forward are counted and reported."""
    from object_detector_amd import weights as W
    from object_detector_amd.trainer import Trainer
    from oracle.train_ref import TorchDetector
    params = W.random_init(2)
    x = onet.synthetic_images(B, S, seed=0)
    anns = _bench_annotations(B, S, seed=1000)
    LS = 1024.0
    tr = Trainer(params, B, (S, S), device=cuda, lr=0.0, momentum=0.9, loss_scale=LS)
    y, npos, _ = tr.pb.encode_batch(anns, return_device=True)
    pred = tr.forward(torch.from_numpy(x).to(cuda)).clone()
    losses = tr.loss(y).clone()
    grads = tr.backward().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(grads).all(), "non-finite gradient at loss scale 1024"
    masks = _device_slope_masks(tr)
    t0 = time.perf_counter()
    ref = TorchDetector(params, dtype=torch.float32, slope_masks=masks)
    ref_losses, ref_grads, ref_pred = ref.loss_and_grads(x, y.cpu().numpy())
    t_oracle = time.perf_counter() - t0
    flips, units = sum(ref.flips.values()), sum(ref.units.values())
    scale = max(1.0, float(np.abs(ref_pred).max()))
    perr = float(np.abs(pred.cpu().numpy() - ref_pred).max())
    print(f"[{B}x{S}] train fwd: max|dlogit| = {perr:.3e} ({perr / scale:.3e} of scale {scale:.2f}); "
          f"kink flips {flips}/{units} = {flips / units:.2e}; oracle {t_oracle:.0f} s; npos {int(npos.sum())}")
    assert perr <= 3e-2 * scale
    np.testing.assert_allclose(losses.cpu().numpy(), ref_losses, rtol=3e-2)
    g = grads.cpu().numpy() / LS
    worst = {}
    for (name, kind), (o, n) in tr.seg.items():
        worst[f"{name}.{kind}"] = _rel(g[o:o + n], ref_grads[f"{name}.{kind}"].reshape(-1))
    top = sorted(worst.items(), key=lambda kv: -kv[1])
    print("worst relative gradient errors:", top[:6], "median", float(np.median(list(worst.values()))))
    _record(f"train_{S}x{B}", dict(max_abs_dlogit=perr, logit_scale=scale, losses=[float(v) for v in losses.cpu().numpy()],
                                   ref_losses=[float(v) for v in ref_losses], kink_flips=flips, leaky_units=units,
                                   worst_grad_rel=top[:6], median_grad_rel=float(np.median(list(worst.values()))),
                                   oracle_seconds=round(t_oracle, 1), oracle_dtype="f32"))
    assert flips / units < 0.02
    assert top[0][1] < 0.01, top[:6]  # measured 0.53 % (32 x 320^2) / 0.65 % (16 x 640^2) against the f32 oracle
    # and the whole step (all-reduce no-op at world 1, SGD, re-pack) runs at this size and leaves finite parameters
    tr.lr = 1e-3
    tr.step(torch.from_numpy(x).to(cuda), anns)
    torch.cuda.synchronize()
    assert torch.isfinite(tr.params).all() and torch.isfinite(tr.losses).all()


def test_load_voc_round_trip_validate_defaults(cuda, tmp_path):
    """A1 through its real entry: weights.save -> ObjectDetector.load_voc(batch_size, input_size, keep_aspect, strict_nms,
    use_multi_gpu) with voc_validate.py's defaults (320x320, batch 16; reference voc_validate.py:14-15,25-27) -> predict on
    images: identical kept indices, classes, confidences and boxes as the detector built from the in-memory parameters,
    and logits within tolerance of the oracle."""
    from object_detector_amd import weights as W
    from object_detector_amd.detector import ObjectDetector
    from object_detector_amd.priors import DEFAULT_PRIOR_WH
    B, S = 16, 320
    params = W.random_init(2)
    path = tmp_path / "voc07+12_320x320.npz"
    W.save(path, params, meta={"prior_wh": np.asarray(DEFAULT_PRIOR_WH)})
    od = ObjectDetector.load_voc(B, input_size=(S, S), keep_aspect=False, strict_nms=False, use_multi_gpu=True,
                                 weights=str(path))
    x = onet.synthetic_images(20, S, seed=7)  # 20 images = one full batch + a partial one
    preds = od.predict(list(x), conf_threshold=0.01)
    ref_od = ObjectDetector(params, B, (S, S), use_multi_gpu=False, device=cuda)
    ref_preds = ref_od.predict(list(x), conf_threshold=0.01)
    assert len(preds) == len(ref_preds) == 20
    for a, b in zip(preds, ref_preds):
        assert np.array_equal(a.flat_indices, b.flat_indices) and np.array_equal(a.classes, b.classes)
        assert np.array_equal(a.confs, b.confs) and np.array_equal(a.bboxes, b.bboxes)
    keep, cnt = od.predict_batch_device(torch.from_numpy(x[:B]).to(cuda), conf_threshold=0.01)
    torch.cuda.synchronize()
    got = od.net.pred.cpu().numpy()
    ref = onet.Runner(params, storage="f16").forward(x[:B])
    _assert_logits(_logit_stats(got, ref, onet.Runner(params, storage="f32").forward(x[:B])), "load_voc 16x320")
    conf, boxes = od.post.conf.cpu().numpy(), od.post.boxes.cpu().numpy()
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for b in range(B):
        r, *_ = onms.detect_image(conf[b], boxes[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
        assert cnt[b] == len(r) and (keep[b, :len(r)] == r).all()
        assert np.array_equal(preds[b].flat_indices, r)


def test_voc_validate_on_a_voc07_layout_directory(cuda, tmp_path):
    """BASELINE.json configs[0]: `voc_validate.py` on 10 VOC07 images at 320x320 -- the reference's acceptance entry point
    (voc_validate.py:9-31) with its own defaults, reading a VOCdevkit-layout directory (JPEGImages / Annotations /
    ImageSets/Main/test.txt; synthetic images of mixed sizes, since the dataset is not available offline) and a weights file
    through `load_voc`.  The logged mAP must equal what the HOST pipeline computes for the same images with the CPU oracle's NMS
    fed the device's conf / boxes: JPEG decode + resize, network, decode, NMS, box mapping and the VOC evaluator are all on
    the path."""
    import subprocess
    import sys
    from PIL import Image
    from object_detector_amd import weights as W
    from object_detector_amd.detector import ObjectDetector, ObjectsPrediction, load_image
    from object_detector_amd.priors import DEFAULT_PRIOR_WH
    import pytoolkit as tk
    rng = np.random.default_rng(42)
    base = tmp_path / "VOCdevkit" / "VOC2007"
    for d in ("Annotations", "JPEGImages", "ImageSets/Main"):
        (base / d).mkdir(parents=True, exist_ok=True)
    ids = []
    for i in range(10):
        h, w = int(rng.integers(200, 500)), int(rng.integers(200, 500))
        low = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)  # blocky, so JPEG keeps some structure
        img = np.repeat(np.repeat(low, 8, 0), 8, 1)[:h, :w]
        name = f"{i:06d}"
        ids.append(name)
        Image.fromarray(img).save(base / "JPEGImages" / f"{name}.jpg", quality=90)
        objs = ""
        for _ in range(int(rng.integers(1, 4))):
            x1, y1 = int(rng.integers(1, w // 2)), int(rng.integers(1, h // 2))
            x2, y2 = int(rng.integers(x1 + 20, w)), int(rng.integers(y1 + 20, h))
            cls = tk.data.voc.CLASS_NAMES[int(rng.integers(0, 20))]
            objs += (f"<object><name>{cls}</name><difficult>0</difficult><bndbox><xmin>{x1}</xmin><ymin>{y1}</ymin>"
                     f"<xmax>{x2}</xmax><ymax>{y2}</ymax></bndbox></object>\n")
        (base / "Annotations" / f"{name}.xml").write_text(
            f"<annotation><filename>{name}.jpg</filename><size><width>{w}</width><height>{h}</height><depth>3</depth></size>\n"
            f"{objs}</annotation>")
    (base / "ImageSets" / "Main" / "test.txt").write_text("\n".join(ids))
    params = W.random_init(2)
    wpath = tmp_path / "voc.npz"
    W.save(wpath, params, meta={"prior_wh": np.asarray(DEFAULT_PRIOR_WH)})
    res = tmp_path / "results"
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "voc_validate.py"), "--vocdevkit-dir", str(tmp_path / "VOCdevkit"),
                        "--result-dir", str(res), "--weights", str(wpath)], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in (res / "validate.log").read_text().splitlines() if "mAP=" in ln]
    assert len(line) == 1
    # the same through the API, with the oracle's NMS on the device's conf / boxes
    X, y = tk.data.voc.load_07_test(tmp_path / "VOCdevkit")
    assert len(X) == 10
    od = ObjectDetector.load_voc(16, input_size=(320, 320), keep_aspect=False, strict_nms=False, use_multi_gpu=False,
                                 weights=str(wpath))
    batch = np.zeros((16, 320, 320, 3), np.uint8)
    for i, pth in enumerate(X):
        batch[i] = load_image(pth, (320, 320), False)
    keep, cnt = od.predict_batch_device(torch.from_numpy(batch).to(cuda), conf_threshold=0.01)
    torch.cuda.synchronize()
    conf, boxes = od.post.conf.cpu().numpy(), od.post.boxes.cpu().numpy()
    preds = []
    for b in range(10):
        k, *_ = onms.detect_image(conf[b], boxes[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
        assert cnt[b].item() == len(k) and (keep[b, :len(k)].cpu().numpy() == k).all()
        preds.append(ObjectsPrediction(k % 20, conf[b].reshape(-1)[k], boxes[b][k // 20], k))
    s = tk.data.voc.evaluate(y, preds)
    want = f'mAP={s["mAP"] * 100:.1f} mAP(VOC2007)={s["mAP_VOC"] * 100:.1f}'
    assert want in line[0], (want, line[0])
    api = od.predict(list(X))
    for a, b in zip(api, preds):
        assert np.array_equal(a.flat_indices, b.flat_indices) and np.array_equal(a.bboxes, b.bboxes)
