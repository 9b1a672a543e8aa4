#!/usr/bin/env python3
# Adapted from ak110/object_detector check_assign.py (MIT): the same argparse flags and tk.* call sequence.
"""Visual check of the prior-box assignment (equivalent of the reference's check_assign.py)."""
import argparse
import pathlib

import numpy as np

import _common  # noqa: F401
import pytoolkit as tk


def _main():
    p = argparse.ArgumentParser()
    p.add_argument("--vocdevkit-dir", default=pathlib.Path("data/VOCdevkit"), type=pathlib.Path)
    p.add_argument("--save-dir", default=pathlib.Path("___assign_check"), type=pathlib.Path)
    p.add_argument("--weights", default=None, type=pathlib.Path)
    p.add_argument("--synthetic", default=0, type=int)
    p.add_argument("--batches", default=32, type=int)
    args = p.parse_args()
    if args.synthetic:
        X_val, y_val = _common.synthetic_dataset(1)
    else:
        X_val, y_val = tk.data.voc.load_07_test(args.vocdevkit_dir)
    X_val, y_val = X_val[:1], y_val[:1]
    # as the reference does (check_assign.py:19,21): the detector at ITS default input size, the generator at 512 x 512 --
    # prior boxes live in normalised image coordinates, so encode_truth / decode_locs do not depend on the image size
    od = _common.make_detector(tk, args, 1, use_multi_gpu=False)
    gen = tk.dl.od.od_gen.create_generator((512, 512), preprocess_input=lambda x: x, encode_truth=od.pb.encode_truth)
    g, _ = gen.flow(X_val, y_val, data_augmentation=True)
    for i, (X_batch, y_batch) in zip(tk.tqdm(range(args.batches)), g):
        for rgb, y in zip(X_batch, y_batch):
            obj_pb = y[:, 1] == 1  # every assigned prior
            classes = np.argmax(y[obj_pb, 2:-4], axis=-1)
            bboxes = od.pb.decode_locs(np.zeros((len(y), 4)), xp=np)[obj_pb, :]  # the prior boxes themselves
            img = tk.ml.plot_objects(rgb, classes, None, bboxes, tk.data.voc.CLASS_NAMES)
            tk.ndimage.save(args.save_dir / f"{i}.jpg", img)


if __name__ == "__main__":
    _main()
