"""GPU parity for K9 (assignment / encode_truth) and K10 (loss fwd+bwd) through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import assign as oassign
from oracle import loss as oloss
from oracle import postprocess as opp

pytestmark = pytest.mark.gpu


def _synthetic_gt(rng, n):
    """SURVEY.md §8d GT recipe."""
    c = rng.uniform(0, 1, (n, 2))
    wh = np.exp(rng.uniform(np.log(0.05), np.log(0.9), (n, 2)))
    b = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)
    return b, rng.integers(0, 20, n).astype(np.int32)


def _annotations(seed, B):
    from object_detector_amd.pb import ObjectsAnnotation
    rng = np.random.default_rng(seed)
    anns = []
    for i in range(B):
        n = 0 if i == 1 else int(np.clip(1 + rng.poisson(1.5), 1, 10))  # one image with NO objects
        b, c = _synthetic_gt(rng, n)
        anns.append(ObjectsAnnotation(None, 320, 320, c, b))
    return anns


@pytest.mark.parametrize("size", [(320, 320), (512, 512)])
def test_encode_truth_bit_exact(cuda, size):
    from object_detector_amd.pb import PriorBoxes
    pb = PriorBoxes(size, 20, device=cuda)
    anns = _annotations(1, 6)
    y, npos, assigned = pb.encode_batch(anns)
    for i, a in enumerate(anns):
        ry, ra = oassign.encode_truth(a.bboxes, a.classes, pb.pb_locs, 20)
        assert (assigned[i] == ra).all()
        assert (y[i] == ry).all()          # targets bit-exact (IEEE f32 div, no FMA)
        assert npos[i] == (ra >= 0).sum()
    # reference check_assign.py:25-27 usage
    obj_pb = y[0][:, 1] == 1
    classes = np.argmax(y[0][obj_pb, 2:-4], axis=-1)
    assert set(classes) <= set(anns[0].classes)
    bboxes = pb.decode_locs(np.zeros((len(y[0]), 4)), xp=np)[obj_pb, :]
    assert (bboxes == pb.pb_locs[obj_pb]).all()
    # decode(encode) round trip gives the GT box back
    dec = pb.decode_locs(y[0][:, -4:], xp=np)[obj_pb]
    gt = anns[0].bboxes[assigned[0][obj_pb]]
    np.testing.assert_allclose(dec, gt, atol=2e-6)


def test_encode_truth_duplicate_and_identical_gt(cuda):
    """ties: two identical GT boxes -> lowest g wins per prior, later g wins the forced prior."""
    from object_detector_amd.pb import ObjectsAnnotation, PriorBoxes
    pb = PriorBoxes((320, 320), 20, device=cuda)
    b = np.array([[0.2, 0.2, 0.6, 0.7], [0.2, 0.2, 0.6, 0.7], [0.0, 0.0, 0.01, 0.01]], np.float32)
    a = ObjectsAnnotation(None, 320, 320, [1, 2, 3], b)
    y, npos, assigned = pb.encode_batch([a])
    ry, ra = oassign.encode_truth(b, [1, 2, 3], pb.pb_locs, 20)
    assert (assigned[0] == ra).all() and (y[0] == ry).all()


@pytest.mark.parametrize("box_mode", ["smooth_l1", "mse"])
def test_loss_matches_oracle(cuda, box_mode):
    from object_detector_amd import ops
    from object_detector_amd.pb import PriorBoxes
    pb = PriorBoxes((320, 320), 20, device=cuda)
    anns = _annotations(2, 4)
    y, npos, assigned = pb.encode_batch(anns, return_device=True)
    rng = np.random.default_rng(4)
    pred = rng.normal(0, 1.5, tuple(y.shape)).astype(np.float32)
    losses, grad = ops.loss_fwd_bwd(torch.from_numpy(pred).to(cuda), y, 20, box_mode=box_mode)
    torch.cuda.synchronize()
    rl, rg = oloss.loss_and_grad(pred, y.cpu().numpy(), 20, box_mode=box_mode)
    np.testing.assert_allclose(losses.cpu().numpy(), rl, rtol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), rg, rtol=1e-4, atol=1e-8)


def test_loss_no_positives(cuda):
    from object_detector_amd import ops
    P = 16800
    y = torch.zeros((2, P, 26), device=cuda)
    y[..., 0] = 1
    pred = torch.randn((2, P, 26), device=cuda)
    losses, grad = ops.loss_fwd_bwd(pred, y)
    rl, rg = oloss.loss_and_grad(pred.cpu().numpy(), y.cpu().numpy())
    np.testing.assert_allclose(losses.cpu().numpy(), rl, rtol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), rg, rtol=1e-4, atol=1e-8)


def test_augment_batch_bit_exact(cuda):
    """K14 vs oracle/augment.py: bit-exact uint8 output for crop/flip/colour/erase on ragged source sizes."""
    from object_detector_amd import od_gen
    from oracle import augment as oaug
    rng = np.random.default_rng(12)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(375, 500), (96, 64), (33, 47), (512, 512)]]
    prm = []
    for i in range(4):
        p = od_gen.AugParams()
        if i != 3:
            p.crop = (0.05 * i, 0.1, 0.9, 1.0 - 0.07 * i)
            p.flip = bool(i % 2)
            p.brightness, p.contrast, p.saturation = 10.0 * i - 12, 0.8 + 0.2 * i, 1.3 - 0.25 * i
            p.erase = [((0.1, 0.2, 0.4, 0.5), (1, 2, 3)), ((0.5, 0.5, 0.95, 0.9), (200, 100, 50))][:i + 1]
        prm.append(p)
    out = od_gen.apply_pixels_device(imgs, prm, (128, 160), cuda).cpu().numpy()
    for i in range(4):
        ref = oaug.augment(imgs[i], (128, 160), prm[i].crop, prm[i].flip, prm[i].brightness, prm[i].contrast,
                           prm[i].saturation, prm[i].erase)
        assert (out[i] == ref).all(), (i, np.abs(out[i].astype(int) - ref.astype(int)).max())
    # identity parameters on a same-size image reproduce the image
    same = od_gen.apply_pixels_device([imgs[3]], [od_gen.AugParams()], (512, 512), cuda).cpu().numpy()[0]
    assert (same == imgs[3]).all()


def test_generator_device_path(cuda, tmp_path):
    """reference check_assign.py:19-27 flow with the device generator + device encode_truth"""
    from object_detector_amd import od_gen
    from object_detector_amd.pb import ObjectsAnnotation, PriorBoxes
    rng = np.random.default_rng(0)
    X = np.array([rng.integers(0, 256, (120, 200, 3), dtype=np.uint8) for _ in range(3)], dtype=object)
    y = np.array([ObjectsAnnotation(None, 200, 120, [i], [[0.2, 0.2, 0.7, 0.8]]) for i in range(3)], dtype=object)
    pb = PriorBoxes((128, 128), 20, device=cuda)
    gen = od_gen.create_generator((128, 128), preprocess_input=lambda x: x, encode_truth=pb.encode_truth, device=cuda)
    g, steps = gen.flow(X, y, batch_size=2, data_augmentation=True, seed=1)
    assert steps == 2
    for _i, (xb, yb) in zip(range(3), g):
        assert xb.dtype == np.uint8 and xb.shape[1:] == (128, 128, 3)
        assert yb.shape[1:] == (len(pb), 26)
        for yy in yb:
            obj = yy[:, 1] == 1
            assert obj.sum() >= 1 and (np.argmax(yy[obj, 2:-4], -1) < 3).all()


def test_generator_feeds_trainer_without_leaving_the_device(cuda):
    """SURVEY.md §8f rank 3 (reference check_generator.py:17-22, docs/MODEL.md:60-64): with on_device=True the generator's
    batches are uint8 DEVICE tensors and encode_truth_device keeps the targets on the device, so a training step consumes
    them as they are -- the pixels and targets never visit the host.  Same seed => same pixels / targets as the numpy path."""
    import torch
    from object_detector_amd import od_gen, weights as W
    from object_detector_amd.pb import ObjectsAnnotation
    from object_detector_amd.trainer import Trainer
    rng = np.random.default_rng(0)
    S, B = 96, 2
    X = np.array([rng.integers(0, 256, (120, 200, 3), dtype=np.uint8) for _ in range(4)], dtype=object)
    y = np.array([ObjectsAnnotation(None, 200, 120, [i], [[0.2, 0.2, 0.7, 0.8]]) for i in range(4)], dtype=object)
    tr = Trainer(W.random_init(2), B, (S, S), device=cuda, lr=0.01, momentum=0.0, loss_scale=256.0)
    gen = od_gen.create_generator((S, S), preprocess_input=None, encode_truth=tr.pb.encode_truth_device, device=cuda,
                                  on_device=True)
    host = od_gen.create_generator((S, S), preprocess_input=None, encode_truth=tr.pb.encode_truth, device=cuda)
    g, steps = gen.flow(X, y, batch_size=B, data_augmentation=True, seed=3)
    gh, _ = host.flow(X, y, batch_size=B, data_augmentation=True, seed=3)
    losses = []
    for _i, (xb, yb), (xh, yh) in zip(range(4), g, gh):
        assert isinstance(xb, torch.Tensor) and xb.is_cuda and xb.dtype == torch.uint8 and tuple(xb.shape) == (B, S, S, 3)
        assert isinstance(yb, torch.Tensor) and yb.is_cuda and tuple(yb.shape) == (B, tr.P, 26)
        assert np.array_equal(xb.cpu().numpy(), xh) and np.array_equal(yb.cpu().numpy(), yh)
        losses.append(float(tr.step(xb, y_target=yb)[3]))
    assert all(np.isfinite(losses)) and tr.skipped_steps == 0
    with pytest.raises(ValueError):
        od_gen.create_generator((S, S), on_device=True)


def test_generator_prefetch_thread_yields_the_same_batches(cuda):
    """flow(..., prefetch=N): the batches come from a background thread on its own HIP stream, N ahead -- the same pixels and
    targets in the same order as the in-thread generator, consumable by the trainer while the thread already builds the next."""
    import torch
    from object_detector_amd import od_gen, weights as W
    from object_detector_amd.pb import ObjectsAnnotation
    from object_detector_amd.trainer import Trainer
    rng = np.random.default_rng(0)
    S, B = 96, 2
    X = np.array([rng.integers(0, 256, (120, 200, 3), dtype=np.uint8) for _ in range(6)], dtype=object)
    y = np.array([ObjectsAnnotation(None, 200, 120, [i % 20], [[0.2, 0.2, 0.7, 0.8]]) for i in range(6)], dtype=object)
    tr = Trainer(W.random_init(2), B, (S, S), device=cuda, lr=0.01, momentum=0.0, loss_scale=256.0)
    gen = od_gen.create_generator((S, S), preprocess_input=None, encode_truth=tr.pb.encode_truth_device, device=cuda,
                                  on_device=True)
    plain, _ = gen.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=5)
    ahead, _ = gen.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=5, prefetch=3)
    losses = []
    for _i, (xa, ya), (xp, yp) in zip(range(7), plain, ahead):
        assert xp.is_cuda and yp.is_cuda
        losses.append(float(tr.step(xp, y_target=yp)[3]))  # consumed while the thread is already three batches further
        assert torch.equal(xa, xp) and torch.equal(ya, yp)
    assert all(np.isfinite(losses))
    ahead.close()  # stops the thread
    with pytest.raises(ValueError):
        od_gen.create_generator((S, S), device=cuda).flow(X, y, batch_size=B, prefetch=2)


def test_generator_device_resident_image_cache(cuda):
    """device_cache=True: every image is decoded / uploaded once and then read by od_augment_batch where it lives in HBM
    (images of different sizes in separate allocations: one base pointer + a 64-bit offset per image).  Same batches, bit for
    bit, as the host-packed path over two epochs; the second epoch loads nothing."""
    import torch
    from object_detector_amd import od_gen
    from object_detector_amd.pb import ObjectsAnnotation, PriorBoxes
    rng = np.random.default_rng(1)
    S, B = 96, 3
    sizes = [(120, 200), (96, 64), (33, 47), (200, 120), (64, 64), (150, 170)]
    X = np.array([rng.integers(0, 256, hw + (3,), dtype=np.uint8) for hw in sizes], dtype=object)
    y = np.array([ObjectsAnnotation(None, hw[1], hw[0], [i], [[0.2, 0.2, 0.7, 0.8]]) for i, hw in enumerate(sizes)], dtype=object)
    pb = PriorBoxes((S, S), 20, device=cuda)
    plain = od_gen.create_generator((S, S), encode_truth=pb.encode_truth_device, device=cuda, on_device=True)
    cached = od_gen.create_generator((S, S), encode_truth=pb.encode_truth_device, device=cuda, on_device=True, device_cache=True)
    loads = []
    orig = cached._load
    cached._load = lambda x: (loads.append(1), orig(x))[1]
    ga, _ = plain.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=9)
    gb, _ = cached.flow(X, y, batch_size=B, data_augmentation=True, shuffle=True, seed=9)
    for _i, (xa, ya), (xb, yb) in zip(range(4), ga, gb):  # two epochs of two batches
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    assert len(loads) == len(X), "every image decoded exactly once"
    with pytest.raises(ValueError):
        od_gen.create_generator((S, S), device_cache=True)
