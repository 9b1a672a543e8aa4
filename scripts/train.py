#!/usr/bin/env python3
"""Train the detector and write the weights file `voc_validate.py --weights` / `ObjectDetector.load_voc` read.

The reference trains through the unseen `fit` of tk.dl.od.ObjectDetector (generator -> encode_truth -> losses:
check_assign.py:21-22, docs/MODEL.md:33-52, per-layer learning rates docs/MODEL.md:84-90) and is accepted by the mAP that
voc_validate.py then logs (README.md:17,22).  This is that loop on MI355X: od_gen generator (pixels augmented on the
device) -> Trainer.step (every tensor op a libodhip.so kernel) -> weights.save.

Data: a VOCdevkit directory (--vocdevkit-dir, image set --image-set of VOC<year>), or --shapes N generated images
(_common.shapes_dataset; VOC07+12 cannot be fetched offline).  Under torch.distributed.run every rank trains on its shard of
the images and the gradients are all-reduced through RCCL (od_allreduce)."""
import argparse
import pathlib
import time

import numpy as np

import _common  # noqa: F401
import pytoolkit as tk


def _main():
    tk.better_exceptions()
    p = argparse.ArgumentParser()
    p.add_argument("--vocdevkit-dir", default=None, type=pathlib.Path)
    p.add_argument("--image-set", default="trainval")
    p.add_argument("--year", default=2007, type=int)
    p.add_argument("--shapes", default=0, type=int, help="N generated images instead of a VOC directory")
    p.add_argument("--result-dir", default=pathlib.Path("results"), type=pathlib.Path)
    p.add_argument("--out", default=None, type=pathlib.Path, help="weights file to write (default <result-dir>/trained.npz)")
    p.add_argument("--init", default=None, type=pathlib.Path, help="start from this weights file (e.g. an imported Darknet53)")
    p.add_argument("--input-size", default=(320, 320), type=int, nargs=2)
    p.add_argument("--batch-size", default=32, type=int, help="per rank")
    p.add_argument("--steps", default=600, type=int)
    p.add_argument("--lr", default=None, type=float,
                   help="peak learning rate (default 0.02 x global batch / 32: the linear scaling rule)")
    p.add_argument("--warmup", default=None, type=int,
                   help="linear warm-up steps (default 50 at a global batch of 32, 200 below it: measured -- 16 x 640^2 from "
                        "scratch diverges with 50 warm-up steps at lr 0.01-0.02 and trains with 200-300)")
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--weight-decay", default=1e-4, type=float)
    p.add_argument("--from-scratch", action="store_true",
                   help="base network at the full learning rate instead of docs/MODEL.md:84-90's 1/100 (no pre-trained "
                        "Darknet53 is available offline)")
    p.add_argument("--box-loss", default="smooth_l1", choices=("smooth_l1", "mse"),
                   help="bounding-box loss: smooth-L1 (north_star) or the mean squared error docs/MODEL.md:46-52 describes")
    p.add_argument("--fit-priors", action="store_true",
                   help="prior-box sizes from KMeans over the training boxes in grid-cell units (docs/MODEL.md:29-31) instead of "
                        "the frozen default table; they travel with the weights file")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--log-every", default=50, type=int)
    p.add_argument("--no-device-cache", action="store_true",
                   help="decode and upload every image every epoch instead of keeping the decoded dataset in HBM")
    p.add_argument("--prefetch", default=2, type=int, help="batches the generator thread runs ahead (0 = in the training thread)")
    args = p.parse_args()
    with tk.dl.session():
        tk.log.init(args.result_dir / "train.log")
        _run(args)


def init_for_training(params, prior=0.01):
    """Prediction-conv bias so that every prior starts at objectness `prior` (the focal-loss initialisation of the RetinaNet
    paper docs/MODEL.md:37 cites): logit(not-obj) - logit(obj) = log((1 - prior) / prior); class and box biases zero."""
    from object_detector_amd import weights as W
    params = dict(params)
    nc, _, _ = W.infer_arch(params)
    b = np.zeros((W.NUM_PRIORS, nc + 6), np.float32)
    half = 0.5 * np.log((1.0 - prior) / prior)
    b[:, 0], b[:, 1] = half, -half
    params["h.out.bias"] = b.reshape(-1)
    return params


def cosine_schedule(lr, steps, warmup):
    def f(i):
        if i < warmup:
            return lr * (i + 1) / warmup
        return lr * 0.5 * (1.0 + np.cos(np.pi * (i - warmup) / max(1, steps - warmup)))
    return f


@tk.log.trace()
def _run(args):
    import torch
    from object_detector_amd import od_gen, weights as W
    from object_detector_amd.net import Context
    from object_detector_amd.trainer import LR_MULTIPLIERS, Trainer, init_comm
    log = tk.log.get(__name__)
    rank, _local, world = tk.dl.dist_env()
    if args.lr is None:
        args.lr = 0.02 * args.batch_size * world / 32.0
    if args.warmup is None:
        args.warmup = 50 if args.batch_size * world >= 32 else 200
    if args.shapes:
        X, y = _common.shapes_dataset(args.shapes, seed=args.seed)
    else:
        X, y = tk.data.voc.load_set(args.vocdevkit_dir, args.year, args.image_set)
    y_all = y
    X, y = X[rank::world], y[rank::world]  # data-parallel: images sharded by index, no collective on the data path
    n = len(X) // args.batch_size * args.batch_size
    if n == 0:
        raise SystemExit(f"rank {rank}: {len(X)} images < one batch of {args.batch_size}")
    X, y = X[:n], y[:n]
    dev = torch.device("cuda", torch.cuda.current_device())
    if args.init is not None:
        params, _meta = W.load(args.init)
    else:
        params = init_for_training(W.random_init(seed=2 + args.seed))
    comm = None
    if world > 1 and torch.distributed.get_backend() == "nccl":
        comm, _ = init_comm(Context.get(dev))
    mult = {k: v for k, v in LR_MULTIPLIERS.items() if not (args.from_scratch and k == "b.")}
    prior_wh = None
    if args.fit_priors:  # from ALL ranks' boxes: every rank must assign against the same priors
        from object_detector_amd import priors as PR
        prior_wh = PR.fit(np.concatenate([a.bboxes for a in y_all if a.num_objects]), tuple(args.input_size), seed=args.seed)
        log.info(f"fitted prior sizes (grid-cell units), level 0: {np.round(prior_wh[0], 2).tolist()}")
    tr = Trainer(params, args.batch_size, tuple(args.input_size), device=dev, lr=args.lr, momentum=args.momentum,
                 weight_decay=args.weight_decay, comm=comm, world_size=world, lr_multipliers=mult, prior_wh=prior_wh, box_mode=args.box_loss)
    gen = od_gen.create_generator(tuple(args.input_size), preprocess_input=None, encode_truth=tr.pb.encode_truth_device,
                                  device=dev, on_device=True, device_cache=not args.no_device_cache)
    batches, per_epoch = gen.flow(X, y, batch_size=args.batch_size, data_augmentation=True, shuffle=True, seed=args.seed + rank,
                                  prefetch=args.prefetch)
    log.info(f"{n} images on rank {rank} of {world}, {per_epoch} steps per epoch, {args.steps} steps, lr {args.lr}, "
             f"multipliers {mult}")
    t0 = time.perf_counter()
    hist = tr.fit(batches, args.steps, lr_schedule=cosine_schedule(args.lr, args.steps, args.warmup), log_every=args.log_every,
                  log=log.info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    batches.close()  # ends the generator's prefetch thread (a thread still issuing GPU work at interpreter exit aborts)
    k = max(1, min(20, args.steps // 10))
    log.info(f"loss first {k} steps {hist[:k, 3].mean():.4f} -> last {k} steps {hist[-k:, 3].mean():.4f}; "
             f"{args.steps} steps in {dt:.1f} s ({args.steps * args.batch_size * world / dt:.0f} images/s), "
             f"skipped {tr.skipped_steps}, loss scale {tr.loss_scale:g}")
    if tk.dl.is_main_process():
        out = args.out or (args.result_dir / "trained.npz")
        out.parent.mkdir(parents=True, exist_ok=True)
        W.save(out, tr.export_params(), meta={"prior_wh": np.asarray(tr.pb.prior_wh), "loss_history": hist[:, 3]})
        log.info(f"weights written to {out}")


if __name__ == "__main__":
    _main()
