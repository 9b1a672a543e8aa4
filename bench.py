#!/usr/bin/env python3
"""Benchmark of the detection hot path on MI355X: images/sec of one full predict step
(uint8 batch resident in HBM -> Darknet53 + neck + shared prediction module -> confidence/decode -> exact top-K ->
NMS kept indices), the metric and workload BASELINE.json names (configs[1]: 320x320, batch 32, 1 GPU, synthetic
VOC-shaped input, random-init weights of the real architecture).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0).  Multi-GPU = images sharded by batch, no data-path collective (weak scaling).
`roofline` is for the dominant device kernel (by summed time), measured with hipEvents on the launch stream inside this
process; `cpu_baseline` is the CPU oracle ("port": torch-CPU conv + numpy NMS, NOT the reference -- SURVEY.md §8c/d)
on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

PEAK_F16_TFLOPS = 2500.0  # dense f16/bf16 MFMA, MI355X (guides/MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 @320, 16 @640)")
    ap.add_argument("--graph", action="store_true", help="replay the network as a hipGraph")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight on separate HIP streams (1 = every step waits for the previous one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--layers", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="train = BASELINE.json configs[3]/[4]: full training step, batch 32 (320) / 16 (640) per GPU")
    return ap.parse_args()


def cpu_baseline(size, seconds):
    """The CPU oracle on this host's cores, same workload shape, bounded sample."""
    from oracle import network as onet
    from oracle import nms as onms
    from oracle import postprocess as opp
    # threads actually usable: the affinity mask capped by the cgroup CPU quota (a 16-core share of a 256-thread host ran
    # torch's default 128 threads ~6x slower than 16 threads do)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(cores)
    params = onet.init_weights(seed=2)
    runner = onet.Runner(params, storage="f32")
    priors = opp.make_priors((size, size))
    bsz = 4
    x = onet.synthetic_images(bsz, size, seed=0)

    def one():
        pred = runner.forward(x)
        conf, boxes = opp.head_postprocess(pred, priors)
        for b in range(bsz):
            onms.detect_image(conf[b], boxes[b])
    one()  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 64:
            break
    return {"value": round(bsz * n / el, 3), "unit": "images/s", "cores": int(cores), "kind": "port",
            "sample": f"{n} iterations x batch {bsz} at {size}x{size} ({el:.1f} s): torch-CPU fp32 conv forward of the "
                      f"same Darknet53+neck+head + numpy decode/top-k/NMS (CPU restatement, not the reference)"}


def train_bench(a, rank, world, dev):
    """One step = prior-box assignment + forward (BN training mode) + loss + backward + gradient all-reduce (RCCL) + SGD."""
    from object_detector_amd import weights as W
    from object_detector_amd.net import Context
    from object_detector_amd.pb import ObjectsAnnotation
    from object_detector_amd.trainer import Trainer, init_comm
    size = a.size
    batch = a.batch or (32 if size <= 320 else 16)
    comm = None
    if world > 1 and torch.distributed.get_backend() == "nccl":
        comm, _ = init_comm(Context.get(dev))  # RCCL communicator through the C ABI (od_comm_*)
    tr = Trainer(W.random_init(2), batch, (size, size), device=dev, lr=1e-3, momentum=0.9, loss_scale=1024.0, comm=comm,
                 world_size=world)
    rng = np.random.default_rng(1000 + rank)
    x = torch.from_numpy(rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
    anns = []
    for _ in range(batch):
        n = int(np.clip(1 + rng.poisson(1.5), 1, 10))
        c = rng.uniform(0, 1, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.05), np.log(0.9), (n, 2)))
        anns.append(ObjectsAnnotation(None, size, size, rng.integers(0, 20, n),
                                      np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.step(x, anns)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step(x, anns)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        nccl = torch.distributed.get_backend() == "nccl"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if nccl else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = tr.losses.cpu().numpy()
    if rank == 0:
        flops_img = 3.0 * (29.01e9 + 7.2e9) * (size / 320.0) ** 2  # fwd + bwd-data + bwd-weight
        print(json.dumps({
            "metric": "train_images_per_sec", "value": round(world * batch * a.steps / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Darknet53 {size}x{size} training step (assign + fwd + focal/CE/smooth-L1 + bwd + "
                                   f"RCCL all-reduce + SGD), batch {batch} per GPU, synthetic data",
                       "global_batch": world * batch, "input_size": size, "parallelism": f"dp{world}",
                       "loss_total": float(losses[3])},
            "roofline": {"bound": "mfma", "achieved": round(flops_img * world * batch * a.steps / elapsed / 1e12 / world, 2),
                         "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops_img * batch * a.steps / elapsed / 1e12 / PEAK_F16_TFLOPS, 4),
                         "note": "whole-step algorithmic conv flops / step time (not a single kernel)", "traffic": None}}))
    if world > 1:
        torch.distributed.destroy_process_group()


def _workgroups(kernel_name, M, N):
    """Workgroups of one launch, from the tile shape in the kernel's template arguments (0 = unknown)."""
    import re
    m = re.match(r"od_conv_8ph<\d+, (\d+)", kernel_name)
    if m:
        bm, bn = 32 * (4 + int(m.group(1))), 256
    else:
        m = re.match(r"od_conv_igemm<(\d+), (\d+)", kernel_name)
        if not m:
            return 0
        bm, bn = int(m.group(1)), int(m.group(2))
    return -(-M // bm) * -(-N // bn)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus and world > 1:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device(f"cuda:{local_rank % ndev}")  # one rank per GPU (the modulo only matters for 1-GPU rehearsals)
    torch.cuda.set_device(dev)
    backend = os.environ.get("OD_BENCH_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; "gloo" to rehearse on one GPU
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    if a.mode == "train":
        return train_bench(a, rank, world, dev)

    from object_detector_amd.detector import ObjectDetector
    size = a.size
    batch = a.batch or (32 if size <= 320 else 16)
    od = ObjectDetector.synthetic(batch, (size, size), seed=2, device=dev, use_multi_gpu=world > 1,
                                  n_inflight=a.inflight)
    rng = np.random.default_rng(1000 + rank)  # each rank its own shard of synthetic images
    x = torch.from_numpy(rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)

    def step():
        # one pass of the hot path over one batch; with --inflight > 1 the step is queued on the next pipeline's stream and
        # overlaps the tail of the previous steps (every step is complete before the closing synchronize)
        if od.n_inflight > 1:
            return od.submit(x, conf_threshold=0.01, graph=a.graph)
        return od.predict_batch_device(x, conf_threshold=0.01, graph=a.graph)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    keep_count = od.post.keep_count.cpu().numpy()

    # ---- roofline of the dominant kernel: hipEvents around every op of the plan, on the launch stream ----------
    reps = 5
    acc = None
    for _ in range(reps):
        ms, names = od.net.time_ops()
        acc = np.asarray(ms) if acc is None else acc + np.asarray(ms)
    ms = acc / reps
    info = od.net.op_info
    groups = {}
    for t, nm, inf in zip(ms, names, info):
        g = groups.setdefault(nm, dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        g["ms"] += float(t)
        g["flops"] += inf["flops"]
        g["bytes"] += inf["bytes"]
        g["n"] += 1
    dom = max(groups, key=lambda k: groups[k]["ms"])
    g = groups[dom]
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    # how much of the chip one launch of that kernel occupies: with batches in flight the plan deliberately picks tiles by
    # CU x time, so e.g. a stage-4 launch of the 256-row kernel is 100 workgroups on 256 CUs (the rest run other batches)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    fill_t, fill_n = 0.0, 0.0
    for t, nm, inf in zip(ms, names, info):
        if nm != dom:
            continue
        wg = _workgroups(nm, inf["shape"][0], inf["shape"][1])
        f = min(1.0, wg / cus) if wg else 1.0
        fill_t += float(t) * f
        fill_n += f
    cu_fill = fill_n / g["n"]
    achieved_active = g["flops"] / (fill_t * 1e-3) / 1e12 if fill_t > 0 else achieved
    net_ms = float(ms.sum())
    # HBM-side bytes per launch of that kernel from the committed PMC passes (profiles/, rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE with the gfx950 x2 FETCH correction); PMC cannot be collected from inside this process -> null if absent
    traffic = None
    try:
        import pathlib
        pj = pathlib.Path(__file__).resolve().parent / "profiles" / "r01" / "pmc_traffic.json"
        if size == 320 and batch == 32 and pj.exists():
            for kname, rec in json.loads(pj.read_text())["kernels"].items():
                if dom in kname:
                    traffic = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"],
                               "algorithmic_bytes_per_launch": round(g["bytes"] / g["n"]),
                               "source": "profiles/r01/pmc_traffic.json"}
    except Exception:
        traffic = None
    if a.layers and rank == 0:
        for t, nm, inf in zip(ms, names, info):
            tf = inf["flops"] / (t * 1e-3) / 1e12
            gb = inf["bytes"] / (t * 1e-3) / 1e9
            print(f"{inf['name']:12s} M={inf['shape'][0]:8d} N={inf['shape'][1]:5d} K={inf['shape'][2]:5d} "
                  f"{t * 1e3:8.1f} us {tf:8.1f} TF/s {gb:8.0f} GB/s  {nm}", file=sys.stderr)
        print(f"network {net_ms:.3f} ms/batch; step {elapsed / a.steps * 1e3:.3f} ms", file=sys.stderr)
        for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"  {k:40s} n={v['n']:3d} {v['ms']:8.3f} ms {v['flops'] / (v['ms'] * 1e-3) / 1e12:8.1f} TF/s",
                  file=sys.stderr)

    if rank == 0:
        total_images = world * batch * a.steps
        out = {
            "metric": "images_per_sec",
            "value": round(total_images / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"Darknet53 {size}x{size} inference (backbone + neck + shared head + decode + "
                                   f"top-k + NMS), batch {batch} per GPU, synthetic VOC-shaped uint8 input, "
                                   f"random-init weights",
                       "global_batch": world * batch, "input_size": size, "parallelism": f"dp{world}",
                       "graph": bool(a.graph), "batches_in_flight": od.n_inflight,
                       "kept_boxes_rank0_img0": int(keep_count[0])},
            "roofline": {"bound": "mfma", "kernel": dom, "launches_per_step": g["n"],
                         "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16_TFLOPS, 4),
                         "avg_launch_us": round(g["ms"] / g["n"] * 1e3, 2),
                         "cu_fill": round(cu_fill, 3), "achieved_on_occupied_cus": round(achieved_active, 2),
                         "algorithmic_gflop_per_launch": round(g["flops"] / g["n"] / 1e9, 3),
                         "network_ms_per_batch": round(net_ms, 4),
                         "network_tflops": round(sum(i["flops"] for i in info) / (net_ms * 1e-3) / 1e12, 2),
                         "traffic": traffic},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(size, a.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
