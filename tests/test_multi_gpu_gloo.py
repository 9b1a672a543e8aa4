"""CPU, world_size 2, gloo: the N>1 inference path = shard images by index, no data-path collective, host gather
(reference knob: use_multi_gpu=True, voc_validate.py:26).  The per-rank device work is replaced by a stub so that
the sharding / gathering / ordering logic is what is under test."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n, q, tmp):
    """The script-level path of scripts/voc_validate.py / voc_evaluate.py: `with tk.dl.session(): tk.log.init(file)` then
    predict.  session() must create the process group from the launcher's environment (nothing else in the scripts
    does), the log FILE and result images must be written by rank 0 only, and the group must be gone afterwards."""
    import pathlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), OD_DIST_BACKEND="gloo")
    import pytoolkit as tk
    tmp = pathlib.Path(tmp)
    assert not dist.is_initialized()
    with tk.dl.session():
        assert dist.is_initialized() and dist.get_world_size() == world and dist.get_rank() == rank
        assert tk.dl.is_main_process() == (rank == 0)
        tk.log.init(tmp / "validate.log")
        tk.log.get("t").info("rank %d reporting", rank)
        tk.ndimage.save(tmp / "img" / "a.jpg", np.full((8, 8, 3), 10 + rank, np.uint8))
        _sharded_predict(rank, world, n, q)
    assert not dist.is_initialized()  # the scope destroys the group it created


def _sharded_predict(rank, world, n, q):
    if True:
        from object_detector_amd.detector import ObjectsPrediction, dist_info, gather_results, shard_indices
        assert dist_info(True) == (rank, world) and dist_info(False) == (0, 1)
        mine = shard_indices(n, rank, world)
        local = {i: ObjectsPrediction([i % 20] * (i % 3), [0.5] * (i % 3), np.full((i % 3, 4), i, np.float32), None)
                 for i in mine}
        allr = gather_results(local, world)
        assert sorted(allr) == list(range(n))
        ok = all((allr[i].bboxes == i).all() and len(allr[i]) == i % 3 for i in range(n))
        q.put((rank, mine, ok))


def test_sharded_predict_world2(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n, world, port = 11, 2, 29731
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == list(range(0, n, 2)) and res[1][1] == list(range(1, n, 2))
    assert all(r[2] for r in res)
    # every image is processed exactly once
    assert sorted(res[0][1] + res[1][1]) == list(range(n))
    # one log file, written by rank 0 alone; one result image, rank 0's
    log = (tmp_path / "validate.log").read_text()
    assert "rank 0 reporting" in log and "rank 1 reporting" not in log
    from PIL import Image
    assert np.asarray(Image.open(tmp_path / "img" / "a.jpg"))[0, 0, 0] in (9, 10, 11)  # rank 0 wrote 10 (JPEG rounding)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from object_detector_amd.trainer import dp_allreduce_, dp_effective_scale, lr_multiplier
        # each rank holds the loss-scaled gradient sums of ITS shard of the global batch
        rng = np.random.default_rng(100 + rank)
        g = torch.from_numpy(rng.normal(0, 1, 1000).astype(np.float32))
        mine = g.clone()
        dp_allreduce_(g)
        q.put((rank, mine.numpy(), g.numpy(), dp_effective_scale(256.0, world), lr_multiplier("b.down1")))
    finally:
        dist.destroy_process_group()


def test_training_gradient_allreduce_world2():
    """training's one exchange step: sum of the per-rank gradient buffers, then 1/(loss_scale*world) in the optimizer =
    the gradient of the mean loss over the global batch (per-rank normalisation)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29741
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for r in res:
        np.testing.assert_allclose(r[2], total, rtol=1e-6)
        assert r[3] == 1.0 / 512.0 and r[4] == 0.01
    assert (res[0][2] == res[1][2]).all()  # every rank ends with bit-identical gradients -> identical SGD steps
