"""GPU parity of one whole training step (forward in BN-training mode, assignment, loss, backward, SGD) against the
torch-CPU float64 oracle with the same seeded parameters, images and ground truth."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cuda, B=2, S=96, seed=2):
    from object_detector_amd import weights as W
    from object_detector_amd.pb import ObjectsAnnotation
    from oracle import network as onet
    params = W.random_init(seed)
    x = onet.synthetic_images(B, S, seed=0)
    rng = np.random.default_rng(3)
    anns = []
    for _ in range(B):
        n = int(rng.integers(1, 4))
        c = rng.uniform(0.2, 0.8, (n, 2)); wh = rng.uniform(0.15, 0.6, (n, 2))
        anns.append(ObjectsAnnotation(None, S, S, rng.integers(0, 20, n),
                                      np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)))
    return params, x, anns


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("backbone_act", [("elu", 1.0), ("leaky", 0.1)], ids=["smooth", "leaky"])
def test_training_step_matches_torch_oracle(cuda, backbone_act):
    """Gradients of every parameter vs the float64 torch oracle: < 1 % on every segment, for both activations.
    With the backbone's LeakyReLU a unit whose pre-activation is within the f16 pipeline's rounding noise of 0 takes the
    other slope on the device than in float64 (tests/test_oracle_golden.py shows ONE such unit in 1.5 M moving some
    segments by 0.6 % even between f32 and f64).  That is a property of the function, not an arithmetic error of the
    kernels, so the oracle differentiates the device's own sign pattern (`slope_masks`); the moved units are COUNTED and
    bounded, and the unmasked comparison is reported beside it."""
    from object_detector_amd.trainer import Trainer
    from oracle.train_ref import TorchDetector
    B, S = 2, 96
    params, x, anns = _setup(cuda, B, S)
    tr = Trainer(params, B, (S, S), device=cuda, lr=0.0, loss_scale=256.0, backbone_act=backbone_act)
    y, _npos, _ = tr.pb.encode_batch(anns, return_device=True)
    pred = tr.forward(torch.from_numpy(x).to(cuda)).clone()
    losses = tr.loss(y).clone()
    grads = tr.backward()
    torch.cuda.synchronize()
    masks = {n.name: (n.z.float() * n.scale + n.shift > 0).cpu().numpy() for n in tr.nodes if n.act and n.act[0] == "leaky"}
    assert len(masks) == (52 if backbone_act[0] == "leaky" else 0)
    ref = TorchDetector(params, backbone_act=backbone_act, slope_masks=masks)
    ref_losses, ref_grads, ref_pred = ref.loss_and_grads(x, y.cpu().numpy())
    assert np.abs(pred.cpu().numpy() - ref_pred).max() <= 3e-2 * max(1.0, np.abs(ref_pred).max())
    np.testing.assert_allclose(losses.cpu().numpy(), ref_losses, rtol=3e-2)
    g = grads.cpu().numpy() / 256.0

    def worst_vs(rg):
        return {(name, kind): _rel(g[o:o + n], rg[f"{name}.{kind}"].reshape(-1)) for (name, kind), (o, n) in tr.seg.items()}
    worst = worst_vs(ref_grads)
    top = sorted(worst.items(), key=lambda kv: -kv[1])
    print("worst relative gradient errors:", top[:5], "median", np.median(list(worst.values())))
    assert max(worst.values()) < 0.01, top[:5]
    if backbone_act[0] == "leaky":
        flips, units = sum(ref.flips.values()), sum(ref.units.values())
        plain = worst_vs(TorchDetector(params, backbone_act=backbone_act).loss_and_grads(x, y.cpu().numpy())[1])
        print(f"units on the other side of the kink than in float64: {flips} of {units} ({flips / units:.2e}); "
              f"without the mask the worst segment differs by {max(plain.values()):.3f}")
        assert 0 < flips < 0.02 * units
        assert max(plain.values()) < 0.15


def test_sgd_step_lowers_loss_and_exports(cuda):
    from object_detector_amd.detector import ObjectDetector
    from object_detector_amd.trainer import Trainer, lr_multiplier
    assert lr_multiplier("b.s3.0.a") == 0.01 and lr_multiplier("h.out") == pytest.approx(1 / 3) and lr_multiplier("n.lat5") == 1.0
    B, S = 2, 96
    params, x, anns = _setup(cuda, B, S)
    tr = Trainer(params, B, (S, S), device=cuda, lr=0.01, momentum=0.0, loss_scale=256.0)
    xt = torch.from_numpy(x).to(cuda)
    first = float(tr.step(xt, anns)[3])
    for _ in range(8):
        last = float(tr.step(xt, anns)[3])
    assert np.isfinite(last) and last < first, (first, last)
    # exported parameters drive the inference path
    od = ObjectDetector(tr.export_params(), B, (S, S), device=cuda)
    keep, cnt = od.predict_batch_device(xt)
    assert int(cnt.min()) >= 0


def test_multi_tensor_pack_and_sgd_equal_per_layer_forms(cuda):
    """The one-launch weight re-pack (LDS-transposed backward layout) and SGD write exactly what the per-layer forms write."""
    from object_detector_amd.trainer import Trainer
    B, S = 2, 96
    params, x, anns = _setup(cuda, B, S)
    tr = Trainer(params, B, (S, S), device=cuda, lr=0.01, momentum=0.9, loss_scale=256.0)
    tr.step(torch.from_numpy(x).to(cuda), anns)
    tr._repack()
    torch.cuda.synchronize()
    got = {k: (tr.wf[k].clone(), tr.wb[k].clone() if tr.wb.get(k) is not None else None) for k in tr.wf}
    for k in tr.wf:
        tr.wf[k].zero_()
        if tr.wb.get(k) is not None:
            tr.wb[k].zero_()
    tr._repack_per_layer()
    torch.cuda.synchronize()
    for k, (f, b) in got.items():
        assert torch.equal(f, tr.wf[k]), k
        if b is not None:
            assert torch.equal(b, tr.wb[k]), k


def test_rccl_comm_single_rank(cuda):
    """od_comm_* through the C ABI with a 1-rank communicator (the only topology a 1-GPU box offers): the all-reduce is
    the identity and leaves the gradient buffer bit-identical; DP semantics proper are covered by the gloo test."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    ctx = Context.get(cuda)
    n = ctx.lib.od_comm_unique_id_bytes()
    buf = (C.c_ubyte * n)()
    _lib.check(ctx.lib.od_comm_get_unique_id(buf, n))
    h = C.c_void_p()
    _lib.check(ctx.lib.od_comm_init(ctx.handle, 0, 1, buf, C.byref(h)))
    g = torch.randn(100003, device=cuda)
    ref = g.clone()
    _lib.check(ctx.lib.od_allreduce(h, g.data_ptr(), g.numel(), _lib.OD_DT_F32, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    _lib.check(ctx.lib.od_comm_destroy(h))


def test_bucketed_allreduce_overlapped_with_backward(cuda, monkeypatch):
    """Gradient buckets (whole layers, cut from the end of the flat buffer) are all-reduced on the communication stream
    while backward is still running.  With a 1-rank RCCL communicator every all-reduce is the identity, so the step must
    be bit-identical to the step without a communicator; the buckets must tile the flat buffer exactly."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    from object_detector_amd.trainer import Trainer
    B, S = 2, 96
    params, x, anns = _setup(cuda, B, S)
    xt = torch.from_numpy(x).to(cuda)
    ref = Trainer(params, B, (S, S), device=cuda, lr=0.01, momentum=0.9, loss_scale=256.0)
    ref.step(xt, anns)
    torch.cuda.synchronize()

    ctx = Context.get(cuda)
    n = ctx.lib.od_comm_unique_id_bytes()
    buf = (C.c_ubyte * n)()
    _lib.check(ctx.lib.od_comm_get_unique_id(buf, n))
    h = C.c_void_p()
    _lib.check(ctx.lib.od_comm_init(ctx.handle, 0, 1, buf, C.byref(h)))
    monkeypatch.setenv("OD_TRAIN_BUCKET_MB", "8")
    tr = Trainer(params, B, (S, S), device=cuda, lr=0.01, momentum=0.9, loss_scale=256.0, comm=h, world_size=1)
    assert tr.cstream is not None and len(tr._buckets) >= 4
    cover = sorted((lo, hi) for lo, hi, _names in tr._buckets)
    assert cover[0][0] == 0 and cover[-1][1] == tr.n_flat and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert set().union(*(names for _lo, _hi, names in tr._buckets)) == set(tr.specs)
    tr.step(xt, anns)
    torch.cuda.synchronize()
    assert tr._next_bucket == len(tr._buckets)
    # every segment is deterministic (no atomics anywhere since the first layer's weight gradient became a slab sum)
    assert torch.equal(tr.grads, ref.grads) and torch.equal(tr.params, ref.params)
    _lib.check(ctx.lib.od_comm_destroy(h))


def test_nonfinite_gradient_skips_update_and_halves_loss_scale(cuda):
    """one f16 overflow in a loss-scaled dz used to put Inf/NaN into the master weights for good.  Now the flat
    gradient buffer is checked on the device after the all-reduce (od_grad_nonfinite) and od_sgd_step_multi leaves weights
    and momentum untouched; the host halves the loss scale when it sees the flag one step later."""
    from object_detector_amd.trainer import Trainer
    B, S = 2, 96
    params, x, anns = _setup(cuda, B, S)
    tr = Trainer(params, B, (S, S), device=cuda, lr=0.01, momentum=0.9, loss_scale=256.0)
    xt = torch.from_numpy(x).to(cuda)
    tr.step(xt, anns)  # a clean step first: momentum is non-zero afterwards
    torch.cuda.synchronize()
    assert int(tr.nonfinite.item()) == 0
    p0, m0 = tr.params.clone(), tr.mom.clone()
    wf0 = {k: v.clone() for k, v in tr.wf.items()}
    y, _n, _ = tr.pb.encode_batch(anns, return_device=True)
    tr._poll_nonfinite()
    tr.forward(xt)
    tr.loss(y)
    tr.grad_pred[0, 5, 3] = float("inf")  # what an overflowing loss gradient looks like
    tr.backward()
    tr.allreduce()
    tr.sgd()
    torch.cuda.synchronize()
    assert int(tr.nonfinite.item()) == 1 and not bool(torch.isfinite(tr.grads).all())
    assert torch.equal(tr.params, p0) and torch.equal(tr.mom, m0)
    assert all(torch.equal(tr.wf[k], wf0[k]) for k in wf0)
    assert tr.skipped_steps == 0 and tr.loss_scale == 256.0
    tr.step(xt, anns)  # the next step learns about it (no stall: the flag was copied asynchronously) and runs clean
    torch.cuda.synchronize()
    assert tr.skipped_steps == 1 and tr.loss_scale == 128.0
    assert int(tr.nonfinite.item()) == 0 and bool(torch.isfinite(tr.params).all()) and not torch.equal(tr.params, p0)
    # NaN is caught as well, anywhere in the buffer (tail elements included)
    g = torch.zeros(1003, device=cuda)
    flag = torch.zeros(1, dtype=torch.int32, device=cuda)
    from object_detector_amd import _lib
    from object_detector_amd.net import _stream_ptr
    for pos in (None, 0, 511, 1000, 1002):
        g.zero_()
        if pos is not None:
            g[pos] = float("nan") if pos % 2 else float("-inf")
        _lib.check(tr.lib.od_grad_nonfinite(tr.ctx.handle, g.data_ptr(), g.numel(), flag.data_ptr(), _stream_ptr()))
        assert int(flag.item()) == int(pos is not None), pos


def test_bf16_casts_round_to_nearest_even(cuda):
    from object_detector_amd import _lib
    from object_detector_amd.net import Context, _stream_ptr
    ctx = Context.get(cuda)
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.normal(0, 1, 100000), rng.normal(0, 1e-30, 1000), rng.normal(0, 1e30, 1000),
                        [0.0, -0.0, np.inf, -np.inf, np.nan, 1.00390625, 1.01171875, 3.3895314e38]]).astype(np.float32)
    src = torch.from_numpy(v).to(cuda)
    dst = torch.empty(v.size, dtype=torch.bfloat16, device=cuda)
    _lib.check(ctx.lib.od_cast_f32_bf16(ctx.handle, src.data_ptr(), dst.data_ptr(), v.size, _stream_ptr()))
    want = src.to(torch.bfloat16)
    assert torch.equal(dst.view(torch.int16)[:-4], want.view(torch.int16)[:-4])
    assert torch.equal(torch.isnan(dst), torch.isnan(want)) and torch.equal(dst[~torch.isnan(dst)], want[~torch.isnan(want)])
    back = torch.empty(v.size, dtype=torch.float32, device=cuda)
    _lib.check(ctx.lib.od_cast_bf16_f32(ctx.handle, dst.data_ptr(), back.data_ptr(), v.size, _stream_ptr()))
    ok = ~torch.isnan(back)
    assert torch.equal(back[ok], want.float()[ok])


def test_bench_multi_rank_train_block_on_a_one_rank_rccl_group(cuda, tmp_path):
    """bench.py's `train` block (what `--gpus N` appends for N > 1: training step with f32 / bf16 payload, bucketed / single
    collective, od_allreduce alone) through the REAL RCCL path -- backend nccl, od_comm_init / od_comm_count / od_allreduce /
    od_comm_destroy -- with the only topology one GPU offers: a 1-rank group, in a child process (it owns a process group)."""
    import subprocess
    import sys
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    code = (
        "import os, sys, json, argparse, torch\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "import bench\n"
        "from object_detector_amd import weights as W\n"
        "os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')\n"
        "dev = torch.device('cuda:0'); torch.cuda.set_device(dev)\n"
        "torch.distributed.init_process_group('nccl', device_id=dev)\n"
        "a = argparse.Namespace(reps=2)\n"
        "out = bench.multi_rank_train_block(W.random_init(2), a, 0, 1, dev, 'nccl')\n"
        "torch.cuda.synchronize(); torch.distributed.destroy_process_group()\n"
        "print('RESULT ' + json.dumps(out))\n")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert len(line) == 1
    import json
    out = json.loads(line[0][7:])
    assert out["od_comm_count"] == 1 and "RCCL" in out["collective"]
    for key in ("step_f32_bucketed", "step_f32_single", "step_bf16_bucketed", "step_bf16_single"):
        assert out[key]["ms_per_step"] > 0 and out[key]["skipped_steps"] == 0, key
    assert out["step_f32_bucketed"]["collectives_per_step"] >= 4 and out["step_f32_single"]["collectives_per_step"] == 1
    for key, nbytes in (("allreduce_only_f32", 4), ("allreduce_only_bf16", 2)):
        assert out[key]["bytes"] % nbytes == 0 and out[key]["ms"] > 0
    print({k: (v.get("ms_per_step") or v.get("ms")) for k, v in out.items() if isinstance(v, dict)})


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
