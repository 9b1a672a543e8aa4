"""Child process of tests/test_gpu_dp.py: ONE data-parallel rank of a real Trainer on cuda:0 (several ranks share the one GPU
of the test box; the collectives run over gloo, the only backend that lets two ranks share a device).  Not a test module.

usage (environment RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set by the parent):
    python tests/dp_worker.py <out_dir> <payload f32|bf16> <bucket_mb>
"""
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def make_batch(B, S, seed=5):
    """The GLOBAL batch (images + annotations), identical in the parent and in every rank."""
    from object_detector_amd.pb import ObjectsAnnotation
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    anns = []
    for _ in range(B):
        n = int(rng.integers(1, 4))
        c = rng.uniform(0.2, 0.8, (n, 2))
        wh = rng.uniform(0.15, 0.6, (n, 2))
        anns.append(ObjectsAnnotation(None, S, S, rng.integers(0, 20, n),
                                      np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)))
    return x, anns


GLOBAL_B, S, LS, LR = 4, 96, 256.0, 0.01


def main():
    out, payload, bucket_mb = pathlib.Path(sys.argv[1]), sys.argv[2], sys.argv[3]
    os.environ["OD_TRAIN_BUCKET_MB"] = bucket_mb
    os.environ["OD_DIST_BACKEND"] = "gloo"
    import torch
    import pytoolkit as tk
    from object_detector_amd import weights as W
    from object_detector_amd.trainer import Trainer
    with tk.dl.session():  # creates the gloo group from the launcher environment, pins cuda:0
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        x, anns = make_batch(GLOBAL_B, S)
        per = GLOBAL_B // world
        xs, an = x[rank * per:(rank + 1) * per], anns[rank * per:(rank + 1) * per]
        tr = Trainer(W.random_init(2), per, (S, S), device="cuda:0", lr=LR, momentum=0.9, loss_scale=LS, comm=None,
                     world_size=world, grad_payload=payload)
        assert (tr.cstream is not None) == (float(bucket_mb) > 0)
        tr.step(torch.from_numpy(xs).to("cuda:0"), an)
        torch.cuda.synchronize()
        if tr.cstream is not None:
            assert tr._next_bucket == len(tr._buckets) >= 2
        np.savez(out / f"rank{rank}_{payload}_{bucket_mb}.npz", grads=tr.grads.cpu().numpy(), params=tr.params.cpu().numpy(),
                 mom=tr.mom.cpu().numpy(), nbuckets=len(tr._buckets))


if __name__ == "__main__":
    main()
