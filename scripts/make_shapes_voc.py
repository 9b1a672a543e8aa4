#!/usr/bin/env python3
"""Write N generated "shapes" images (_common.shapes_dataset) as a VOCdevkit-layout directory, so that voc_validate.py /
voc_evaluate.py / train.py can be run end to end without the (offline-unavailable) PASCAL VOC data."""
import argparse
import pathlib

import _common


def _main():
    p = argparse.ArgumentParser()
    p.add_argument("vocdevkit_dir", type=pathlib.Path)
    p.add_argument("-n", default=64, type=int)
    p.add_argument("--seed", default=1000, type=int)
    p.add_argument("--image-set", default="test")
    p.add_argument("--format", default="png", choices=("png", "jpg"))
    args = p.parse_args()
    X, y = _common.shapes_dataset(args.n, seed=args.seed)
    base = _common.write_voc_layout(args.vocdevkit_dir, X, y, image_set=args.image_set, fmt=args.format)
    print(f"{args.n} images -> {base}")


if __name__ == "__main__":
    _main()
