#!/usr/bin/env python3
"""Is the 3-in-flight pipeline bound by chip throughput or by each stream's chain latency?  Post-processing (a ~0.3 ms chain of
small kernels) on a stream of its own per pipeline, so that the pipeline's network stream is free for its next batch as soon
as the network is done.  usage: exp_post_stream.py [size] [batch] [inflight]"""
import pathlib, sys
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent.parent))
import bench
from object_detector_amd import weights as W
from object_detector_amd.detector import ObjectDetector

size = int(sys.argv[1]) if len(sys.argv) > 1 else 320
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if size == 320 else 16)
nin = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
od = ObjectDetector(W.random_init(2), batch, (size, size), device=dev, n_inflight=nin)
x = torch.from_numpy(np.random.default_rng(1000).integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
base = lambda: od.submit(x, conf_threshold=0.01)
for rep in range(2):
    print(f"post on the network stream      {bench._median(bench.timed_reps(base, 30, 5, 5, 1, dev)) / 30 * 1e3:.4f} ms/step", flush=True)
pstreams = [torch.cuda.Stream(device=dev) for _ in od._pipes]
pdone = [torch.cuda.Event() for _ in od._pipes]
nets_done = [torch.cuda.Event() for _ in od._pipes]
state = {"i": 0}
def split():
    i = state["i"]; state["i"] = (i + 1) % len(od._pipes)
    p = od._pipes[i]
    p.stream.wait_event(pdone[i])  # this pipeline's previous post-processing still reads pred
    with torch.cuda.stream(p.stream):
        pred = p.net.forward(x)
        nets_done[i].record()
    pstreams[i].wait_event(nets_done[i])
    with torch.cuda.stream(pstreams[i]):
        p.post.run(pred, 0.01)
        pdone[i].record()
def sync_all():
    for s in pstreams: s.synchronize()
for rep in range(2):
    ts = bench.timed_reps(lambda: split(), 30, 5, 5, 1, dev)
    print(f"post on a stream of its own     {bench._median(ts) / 30 * 1e3:.4f} ms/step", flush=True)
