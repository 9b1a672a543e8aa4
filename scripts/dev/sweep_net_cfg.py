#!/usr/bin/env python3
"""In-network tile-config sweep: time selected layers of the real plan (cold caches, real producers/consumers around
them) under different tile_cfg overrides.  usage: sweep_net_cfg.py --layers "b.s3*a,b.s4*a" --cfgs -1,3,6,7"""
import argparse
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
from object_detector_amd import weights as W  # noqa: E402
from object_detector_amd.net import Net  # noqa: E402


def match(name, pat):
    pre, _, suf = pat.partition("*")
    return name.startswith(pre) and name.endswith(suf)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="b.s3*a,b.s4*a,b.s5*a")
    ap.add_argument("--cfgs", default="-1,3,6,7")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--per-op", action="store_true")
    a = ap.parse_args()
    pats = a.layers.split(",")
    params = W.random_init(2)
    x = torch.randint(0, 256, (a.batch, a.size, a.size, 3), dtype=torch.uint8, device="cuda:0")
    print("cfg   " + "  ".join(f"{p:>10s}" for p in pats) + "    network_ms")
    for cfg in [int(c) for c in a.cfgs.split(",")]:
        try:
            net = Net(params, a.batch, (a.size, a.size), tile_cfg={} if cfg < 0 else {p: cfg for p in pats})
        except Exception as e:  # config does not support the shape
            print(f"{cfg:3d}   unsupported: {str(e)[:80]}")
            continue
        net.forward(x)
        acc = None
        for _ in range(a.reps):
            ms, _names = net.time_ops()
            acc = np.asarray(ms) if acc is None else acc + np.asarray(ms)
        ms = acc / a.reps
        names = [i["name"] for i in net.op_info]
        cols = []
        for p in pats:
            v = [t for t, n in zip(ms, names) if match(n, p)]
            cols.append(f"{np.mean(v) * 1e3:8.1f}us" if v else "       -  ")
        print(f"{cfg:3d}   " + "  ".join(cols) + f"    {ms.sum():.3f}", flush=True)
        if a.per_op:
            for t, i in zip(ms, net.op_info):
                if any(match(i["name"], p) for p in pats):
                    print(f"        {i['name']:10s} M={i['shape'][0]:7d} N={i['shape'][1]:5d} K={i['shape'][2]:5d} {t * 1e3:8.1f} us")
        del net


if __name__ == "__main__":
    main()
