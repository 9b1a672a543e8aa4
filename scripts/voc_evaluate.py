#!/usr/bin/env python3
# Adapted from ak110/object_detector voc_evaluate.py (MIT): the same argparse flags and tk.* call sequence.
"""Per-class precision / recall / F at conf 0.6 + plots of the first 64 predictions (equivalent of the reference's
voc_evaluate.py entry point)."""
import argparse
import pathlib

import _common  # noqa: F401
import pytoolkit as tk


def _main():
    tk.better_exceptions()
    p = argparse.ArgumentParser()
    p.add_argument("--vocdevkit-dir", default=pathlib.Path("data/VOCdevkit"), type=pathlib.Path)
    p.add_argument("--result-dir", default=pathlib.Path("results"), type=pathlib.Path)
    p.add_argument("--input-size", default=(320, 320), type=int, nargs=2)
    p.add_argument("--batch-size", default=16, type=int)
    p.add_argument("--weights", default=None, type=pathlib.Path)
    p.add_argument("--precision", default=None, choices=("f16", "mixed"),
                   help="mixed = every logit within 1e-3 x scale of an fp32 run (ObjectDetector(precision=...)); default f16")
    p.add_argument("--synthetic", default=0, type=int)
    args = p.parse_args()
    with tk.dl.session():
        tk.log.init(args.result_dir / "evaluate.log")
        _run(args)


@tk.log.trace()
def _run(args):
    if args.synthetic:
        X_test, y_test = _common.synthetic_dataset(args.synthetic)
    else:
        X_test, y_test = tk.data.voc.load_07_test(args.vocdevkit_dir)
    od = _common.make_detector(tk, args, args.batch_size, tuple(args.input_size), keep_aspect=False, strict_nms=False,
                               use_multi_gpu=True, precision=args.precision)
    pred = od.predict(X_test, conf_threshold=0.6)
    precisions, recalls, fscores, supports = tk.ml.compute_scores(y_test, pred, iou_threshold=0.5,
                                                                  num_classes=len(tk.data.voc.CLASS_NAMES))
    tk.ml.print_scores(precisions, recalls, fscores, supports, tk.data.voc.CLASS_NAMES,
                       print_fn=tk.log.get(__name__).info)
    save_dir = args.result_dir / "___check"
    for i, (x, pr) in enumerate(zip(X_test[:64], pred[:64])):
        img = pr.plot(x, tk.data.voc.CLASS_NAMES)
        name = x.stem if isinstance(x, pathlib.Path) else f"{i:06d}"
        tk.ndimage.save(save_dir / (name + ".jpg"), img)


if __name__ == "__main__":
    _main()
