#!/usr/bin/env python3
# Adapted from ak110/object_detector voc_validate.py (MIT): the same argparse flags and tk.* call sequence.
"""VOC07-test mAP of the detector (this build's equivalent of the reference's voc_validate.py entry point;
same flags plus --weights / --synthetic N, because VOC data and trained weights are not available offline)."""
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
    p.add_argument("--synthetic", default=0, type=int, help="N synthetic images + random-init weights instead of VOC")
    args = p.parse_args()
    with tk.dl.session():
        tk.log.init(args.result_dir / "validate.log")
        _run(args)


@tk.log.trace()
def _run(args):
    if args.synthetic:
        X_test, y_test = _common.synthetic_dataset(args.synthetic)
    else:
        X_test, y_test = tk.data.voc.load_07_test(args.vocdevkit_dir)
    od = _common.make_detector(tk, args, args.batch_size, tuple(args.input_size), keep_aspect=False, strict_nms=False,
                               use_multi_gpu=True, precision=args.precision)
    pred_test = od.predict(X_test)
    scores = tk.data.voc.evaluate(y_test, pred_test)
    tk.log.get(__name__).info(f'mAP={scores["mAP"] * 100:.1f} mAP(VOC2007)={scores["mAP_VOC"] * 100:.1f}')


if __name__ == "__main__":
    _main()
