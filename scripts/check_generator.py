#!/usr/bin/env python3
# Adapted from ak110/object_detector check_generator.py (MIT): the same argparse flags and tk.* call sequence.
"""Visual check of the data generator / augmentation (equivalent of the reference's check_generator.py)."""
import argparse
import pathlib

import _common  # noqa: F401
import pytoolkit as tk


def _main():
    p = argparse.ArgumentParser()
    p.add_argument("--vocdevkit-dir", default=pathlib.Path("data/VOCdevkit"), type=pathlib.Path)
    p.add_argument("--save-dir", default=pathlib.Path("___generator_check"), type=pathlib.Path)
    p.add_argument("--synthetic", default=0, type=int)
    p.add_argument("--batches", default=32, type=int)
    args = p.parse_args()
    if args.synthetic:
        X_val, y_val = _common.synthetic_dataset(1)
    else:
        X_val, y_val = tk.data.voc.load_07_test(args.vocdevkit_dir)
    X_val, y_val = X_val[:1], y_val[:1]
    gen = tk.dl.od.od_gen.create_generator((512, 512), preprocess_input=lambda x: x, encode_truth=None)
    g, _ = gen.flow(X_val, y_val, data_augmentation=True)
    for i, (X_batch, y_batch) in zip(tk.tqdm(range(args.batches)), g):
        for rgb, y in zip(X_batch, y_batch):
            img = tk.ml.plot_objects(rgb, y.classes, None, y.bboxes, tk.data.voc.CLASS_NAMES)
            tk.ndimage.save(args.save_dir / f"{i}.jpg", img)


if __name__ == "__main__":
    _main()
