#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output dirs: per kernel, mean duration (us) and mean of every counter per dispatch.
usage: python scripts/dev/pmc_summarize.py <dir> [<dir> ...] [--match substr]"""
import collections
import csv
import glob
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    for d in args:
        dur = collections.defaultdict(list)
        ctr = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                ctr[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in dur:
            if match and match not in k:
                continue
            if not match and "od_" not in k:
                continue
            v = dur[k]
            line = f"{d.rstrip('/').split('/')[-1]:14s} {k[:100]:100s} n={len(v):3d} {sum(v) / len(v):9.1f} us"
            for c, vals in sorted(ctr.get(k, {}).items()):
                line += f" | {c}={sum(vals) / len(vals):.4g}"
            print(line)


if __name__ == "__main__":
    main()
