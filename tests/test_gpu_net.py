"""GPU end-to-end parity: the native forward plan (Darknet53 + neck + shared prediction module) and the whole
predict path vs the CPU oracle, same seeded weights and images.

Tolerances (north_star: 1e-3 on logits, bit-exact kept indices):
  * vs the oracle with f16 storage at the SAME rounding points (storage="f16"): the size-independent criteria of
    oracle/compare.py (rms |dlogit| <= 3e-4 of the logit scale, maximum inside the Gaussian tail of that rms, and the
    device as close to the fp32 oracle as the CPU f16-storage run is); at this size (78 k logits) the maximum is
    additionally held to the 1e-3 * scale of round 1, which the full-size tests (14 M logits) cannot meet by statistics
    alone -- see tests/test_gpu_fullsize.py
  * vs the plain fp32-activation oracle: bounded at 2e-2 * scale (f16 activation storage is the design
    point of the path -- DESIGN.md "numerics"); never used to claim index parity
  * kept indices: bit-exact vs oracle NMS fed the device's own (conf, boxes); and equal to the full-oracle result
    whenever no candidate's confidence gap is below the logit tolerance
"""
import numpy as np
import pytest
import torch

from oracle import network as onet
from oracle import nms as onms
from oracle import postprocess as opp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_setup(cuda):
    from object_detector_amd import weights as W
    from object_detector_amd.net import Net
    params = W.random_init(seed=2)
    B, S = 2, 96
    net = Net(params, B, (S, S), device=cuda)
    x = onet.synthetic_images(B, S, seed=0)
    pred = net.forward(torch.from_numpy(x).to(cuda)).clone()
    torch.cuda.synchronize()
    return params, net, x, pred.cpu().numpy()


def test_logits_vs_f16_storage_oracle(small_setup):
    params, net, x, got = small_setup
    from oracle.compare import assert_logits, logit_stats
    ref = onet.Runner(params, storage="f16").forward(x)
    assert got.shape == ref.shape
    rec = logit_stats(got, ref, onet.Runner(params, storage="f32").forward(x))
    assert_logits(rec, "2x96")
    assert rec["max_abs_dlogit"] <= 1.2e-3 * rec["logit_scale"], rec


def test_logits_vs_fp32_oracle(small_setup):
    params, net, x, got = small_setup
    ref = onet.Runner(params, storage="f32").forward(x)
    scale = max(1.0, np.abs(ref).max())
    err = np.abs(got - ref).max()
    print(f"max |logit - fp32 oracle| = {err:.3e} (scale {scale:.2f})")
    assert err <= 2e-2 * scale


def test_graph_replay_matches_eager(small_setup, cuda):
    params, net, x, got = small_setup
    xt = torch.from_numpy(x).to(cuda)
    a = net.forward(xt, graph=True).clone()
    b = net.forward(xt, graph=True).clone()
    torch.cuda.synchronize()
    assert (a.cpu().numpy() == got).all() and (b.cpu().numpy() == got).all()  # split-K slabs are summed in a fixed order


def test_predict_end_to_end(cuda):
    from object_detector_amd.detector import ObjectDetector
    B, S = 2, 160
    od = ObjectDetector.synthetic(B, (S, S), seed=2, device=cuda)
    x = onet.synthetic_images(B, S, seed=0)
    keep, cnt = od.predict_batch_device(torch.from_numpy(x).to(cuda), conf_threshold=0.01)
    torch.cuda.synchronize()
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    conf, boxes = od.post.conf.cpu().numpy(), od.post.boxes.cpu().numpy()
    for b in range(B):
        ref, *_ = onms.detect_image(conf[b], boxes[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
        assert cnt[b] == len(ref) and (keep[b, :len(ref)] == ref).all()
    # public API on arrays: same result, classes/confs/bboxes consistent with the flat indices
    preds = od.predict(list(x), conf_threshold=0.01)
    for b, p in enumerate(preds):
        assert (p.flat_indices == keep[b, :cnt[b]]).all()
        assert (p.classes == p.flat_indices % 20).all()
        assert (p.confs == conf[b].reshape(-1)[p.flat_indices]).all()
        assert (np.diff(p.confs) <= 0).all()
    # conf_threshold=0.6 (voc_evaluate.py:27) on random weights: nothing (or few) survives, still well-formed
    hi = od.predict(list(x), conf_threshold=0.6)
    assert all(len(p) <= 200 and (p.confs > 0.6).all() for p in hi)


def test_entry_points_synthetic(cuda, tmp_path):
    """scripts/voc_validate.py / voc_evaluate.py / check_assign.py equivalents run end to end (synthetic data + weights)."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    for script, extra in (("voc_validate.py", ["--synthetic", "5", "--batch-size", "2", "--input-size", "96", "96"]),
                          ("voc_evaluate.py", ["--synthetic", "3", "--batch-size", "2", "--input-size", "96", "96"]),
                          ("check_assign.py", ["--synthetic", "1", "--batches", "2", "--save-dir", str(tmp_path / "assign")])):
        r = subprocess.run([sys.executable, str(root / "scripts" / script), "--result-dir", str(tmp_path)] + extra
                           if script.startswith("voc") else [sys.executable, str(root / "scripts" / script)] + extra,
                           capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
    assert "mAP=" in (tmp_path / "validate.log").read_text()
    assert len(list((tmp_path / "assign").glob("*.jpg"))) == 2


def test_keep_aspect_boxes_map_back(cuda):
    from object_detector_amd.detector import ObjectDetector, load_image
    img = np.random.default_rng(0).integers(0, 256, (100, 200, 3), dtype=np.uint8)
    canvas, (sx, sy) = load_image(img, (96, 96), keep_aspect=True, return_scale=True)
    assert canvas.shape == (96, 96, 3) and sx == 1.0 and abs(sy - 0.5) < 1e-6 and (canvas[48:] == 0).all()
    od = ObjectDetector.synthetic(2, (96, 96), keep_aspect=True, device=cuda)
    p = od.predict([img, img])
    assert len(p) == 2 and (p[0].bboxes >= 0).all() and (p[0].bboxes <= 1).all()


def test_batches_in_flight_match_sequential(cuda):
    """submit()/collect(): three different batches queued on three pipelines (own buffers, own HIP streams) give exactly the
    kept indices of the one-at-a-time path, and predict() (which pipelines its batches) returns them in input order."""
    from object_detector_amd.detector import ObjectDetector
    od = ObjectDetector.synthetic(2, (64, 64), seed=4, device=cuda, use_multi_gpu=False, n_inflight=3)
    assert od.n_inflight == 3
    rng = np.random.default_rng(0)
    xs = [torch.from_numpy(rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)).to(cuda) for _ in range(5)]
    ref = []
    for x in xs:
        kf, kc = od.predict_batch_device(x, conf_threshold=0.01)
        torch.cuda.synchronize()
        ref.append((kf.cpu().numpy().copy(), kc.cpu().numpy().copy()))
    tickets = [od.submit(x, 0.01) for x in xs[:3]]
    for t, (rkf, rkc) in zip(tickets, ref[:3]):
        kf, kc = od.collect(t)
        assert np.array_equal(kc.cpu().numpy(), rkc) and np.array_equal(kf.cpu().numpy(), rkf)
    # a pipeline is reused in stream order: ticket 0 again
    t = od.submit(xs[3], 0.01)
    kf, kc = od.collect(t)
    assert np.array_equal(kc.cpu().numpy(), ref[3][1]) and np.array_equal(kf.cpu().numpy(), ref[3][0])
    imgs = [x[i].cpu().numpy() for x in xs for i in range(2)]   # 10 images = 5 batches through predict()
    preds = od.predict(imgs, conf_threshold=0.01)
    assert len(preds) == 10
    for b, (rkf, rkc) in enumerate(ref):
        for i in range(2):
            assert np.array_equal(preds[2 * b + i].flat_indices, rkf[i, :rkc[i]])


def test_predict_from_files_with_decode_pool(cuda, tmp_path, monkeypatch):
    """predict() on image files: the decode thread pool + pinned staging buffers return, in input order, exactly what the
    single-threaded path returns (11 files of mixed sizes = 5 full batches + a partial one + buffer rotation)."""
    from PIL import Image
    from object_detector_amd.detector import ObjectDetector
    rng = np.random.default_rng(5)
    paths = []
    for i in range(11):
        h, w = int(rng.integers(60, 200)), int(rng.integers(60, 200))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pth = tmp_path / f"im{i}.png"
        Image.fromarray(a).save(pth)
        paths.append(str(pth))
    od = ObjectDetector.synthetic(2, (96, 96), seed=4, device=cuda, use_multi_gpu=False, n_inflight=3)
    monkeypatch.setenv("OD_DECODE_THREADS", "1")
    one = od.predict(paths, conf_threshold=0.01)
    monkeypatch.setenv("OD_DECODE_THREADS", "8")
    many = od.predict(paths, conf_threshold=0.01)
    again = od.predict(paths[::-1], conf_threshold=0.01)[::-1]
    monkeypatch.setenv("OD_DECODE_PROCS", "2")  # spawned decode workers + shared-memory staging block
    procs = od.predict(paths, conf_threshold=0.01)
    od.close_decode_pool()
    assert len(one) == len(many) == len(procs) == 11
    for a, b, c, d in zip(one, many, again, procs):
        assert np.array_equal(a.flat_indices, b.flat_indices) and np.array_equal(a.bboxes, b.bboxes)
        assert np.array_equal(a.flat_indices, c.flat_indices)
        assert np.array_equal(a.flat_indices, d.flat_indices) and np.array_equal(a.bboxes, d.bboxes)


@pytest.mark.parametrize("case", [(8, 352, 480, "1"), (20, 256, 256, "0"), (12, 512, 384, "1")], ids=str)
def test_shape_driven_kernel_selection_at_odd_sizes(cuda, case, monkeypatch):
    """Which kernel a layer takes depends on its map size (the streaming / weights-resident kernels and the 1x1 layers that
    ride in their producer's launch switch on and off with it): sizes and batch sizes no other test uses, the throughput and
    the latency plan, fused blocks and layer by layer -- logits by the size-independent criteria of oracle/compare.py, kept
    indices bit-exact against the oracle NMS fed the device's conf / boxes."""
    from object_detector_amd.detector import ObjectDetector
    from oracle.compare import assert_logits, logit_stats
    B, H, W, fuse = case
    monkeypatch.setenv("OD_FUSE_BLOCKS", fuse)
    x = np.random.default_rng(B * 1000 + H).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    xt = torch.from_numpy(x).to(cuda)
    ref = ref32 = None
    for nin in (3, 1):
        od = ObjectDetector.synthetic(B, (H, W), seed=2, device=cuda, use_multi_gpu=False, n_inflight=nin)
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
        assert_logits(logit_stats(pred, ref, ref32), f"{B}x{H}x{W} fuse={fuse} inflight {nin}")
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        for b in range(B):
            r, *_ = onms.detect_image(conf[b], boxes[b], K=1024, conf_threshold=0.01, iou_threshold=0.45, max_det=200)
            assert cnt[b] == len(r) and (keep[b, :len(r)] == r).all(), f"image {b}: kept indices differ"
        print(f"{case} inflight {nin}:", [n for n in names if any(k in n for k in ("rdirect", "stream3", "true, false>", "false, true>", "stem", "bneck"))])
        del od
        torch.cuda.empty_cache()
