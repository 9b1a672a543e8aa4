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
    ap.add_argument("--reps", type=int, default=5,
                    help="the K-step timed loop is repeated this many times (each bracketed by barrier + synchronize); the "
                         "line reports the MEDIAN repetition")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 @320, 16 @640)")
    ap.add_argument("--graph", action="store_true", help="replay the network as a hipGraph")
    ap.add_argument("--inflight", type=int, default=3,
                    help="batches in flight on separate HIP streams (1 = every step waits for the previous one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the `extra` block (the other configurations of BASELINE.json's metric: 640x640 batch 16, "
                         "batch 1 at both sizes, the training steps)")
    ap.add_argument("--layers", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="train = BASELINE.json configs[3]/[4]: full training step, batch 32 (320) / 16 (640) per GPU")
    ap.add_argument("--grad-payload", default="f32", choices=["f32", "bf16"],
                    help="train: gradient all-reduce payload (bf16 = BASELINE.json configs[4])")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks OURSELVES (one process per GPU,
    `torch.distributed.run`, rendezvous on 127.0.0.1) and forward their output and exit code.  This parent has not made
    and will not make a GPU call besides counting the devices, and it only ever SPAWNS children (it never replaces itself
    with another program), so nothing that touched the GPU is ever re-executed.  The knob this stands for is the reference's use_multi_gpu=True (voc_validate.py:26)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and "OD_BENCH_BACKEND" not in env:
        # fewer GPUs than ranks (a rehearsal on a 1-GPU box): ranks share devices, which RCCL cannot do -> gloo collectives
        print(f"bench.py: {a.gpus} ranks on {ndev} GPU(s): rehearsal with the gloo backend (ranks share a device)",
              file=sys.stderr)
        env["OD_BENCH_BACKEND"] = "gloo"
    # --standalone: torchrun's own c10d rendezvous picks the port (no bind-then-close race with other processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={a.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def usable_cores():
    """threads actually usable: the affinity mask capped by the cgroup CPU quota (a 16-core share of a 256-thread host ran
    torch's default 128 threads ~6x slower than 16 threads do)"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(size, seconds):
    """The CPU oracle on this host's cores, same workload shape, bounded sample."""
    from oracle import network as onet
    from oracle import nms as onms
    from oracle import postprocess as opp
    cores = usable_cores()
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


def _barrier(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def timed_reps(step, steps, warmup, reps, world, dev):
    """W untimed warm-up steps, then `reps` repetitions of EXACTLY `steps` steps, each bracketed by barrier +
    torch.cuda.synchronize() on both sides; every repetition's time is the MAX over ranks.  -> sorted list of seconds."""
    for _ in range(warmup):
        step()
    out = []
    for _ in range(max(1, reps)):
        _barrier(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        _barrier(world)
        out.append(time.perf_counter() - t0)
    if world > 1:
        nccl = torch.distributed.get_backend() == "nccl"
        t = torch.tensor(out, dtype=torch.float64, device=dev if nccl else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        out = [float(v) for v in t.tolist()]
    return sorted(out)


def _median(xs):
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def bench_annotations(batch, size, rng):
    from object_detector_amd.pb import ObjectsAnnotation
    anns = []
    for _ in range(batch):
        n = int(np.clip(1 + rng.poisson(1.5), 1, 10))
        c = rng.uniform(0, 1, (n, 2))
        wh = np.exp(rng.uniform(np.log(0.05), np.log(0.9), (n, 2)))
        anns.append(ObjectsAnnotation(None, size, size, rng.integers(0, 20, n),
                                      np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)))
    return anns


TRAIN_FLOPS_IMG_320 = 3.0 * (29.01e9 + 7.2e9)  # fwd + bwd-data + bwd-weight of backbone + neck/head, per 320x320 image


def run_train(params, size, batch, steps, warmup, reps, rank, world, dev, comm=None, grad_payload="f32"):
    """One step = prior-box assignment + forward (BN training mode) + loss + backward + gradient all-reduce + SGD."""
    from object_detector_amd.trainer import Trainer
    tr = Trainer(params, batch, (size, size), device=dev, lr=1e-3, momentum=0.9, loss_scale=1024.0, comm=comm,
                 world_size=world, grad_payload=grad_payload)
    rng = np.random.default_rng(1000 + rank)
    x = torch.from_numpy(rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
    anns = bench_annotations(batch, size, rng)
    times = timed_reps(lambda: tr.step(x, anns), steps, warmup, reps, world, dev)
    el = _median(times)
    flops_img = TRAIN_FLOPS_IMG_320 * (size / 320.0) ** 2
    rec = {"value": round(world * batch * steps / el, 2), "ms_per_step": round(el / steps * 1e3, 3),
           "ms_per_step_reps": [round(t / steps * 1e3, 3) for t in times],
           "tflops_per_gpu": round(flops_img * batch * steps / el / 1e12, 2),
           "frac": round(flops_img * batch * steps / el / 1e12 / PEAK_F16_TFLOPS, 4),
           "loss_total": float(tr.losses.cpu().numpy()[3]), "skipped_steps": tr.skipped_steps,
           "gradient_buckets": len(tr._buckets), "grad_payload": tr.grad_payload}
    return rec, tr


def train_bench(a, rank, world, dev):
    from object_detector_amd import _lib, weights as W
    from object_detector_amd.net import Context
    from object_detector_amd.trainer import init_comm
    import ctypes as C
    size = a.size
    batch = a.batch or (32 if size <= 320 else 16)
    comm, seen = None, world
    if world > 1 and torch.distributed.get_backend() == "nccl":
        ctx = Context.get(dev)
        comm, _ = init_comm(ctx)  # RCCL communicator through the C ABI (od_comm_*)
        r_, n_ = C.c_int(), C.c_int()
        _lib.check(ctx.lib.od_comm_count(comm, C.byref(r_), C.byref(n_)), "od_comm_count")
        seen = n_.value
        assert seen == world and r_.value == rank, f"RCCL reports rank {r_.value} of {seen}, launcher said {rank} of {world}"
    elif world > 1:
        seen = torch.distributed.get_world_size()
    rec, _tr = run_train(W.random_init(2), size, batch, a.steps, a.warmup, a.reps, rank, world, dev, comm, a.grad_payload)
    if rank == 0:
        print(json.dumps({
            "metric": "train_images_per_sec", "value": rec["value"], "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "reps": a.reps, "ms_per_step": rec["ms_per_step"],
            "ms_per_step_reps": rec["ms_per_step_reps"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Darknet53 {size}x{size} training step (assign + fwd + focal/CE/smooth-L1 + bwd + "
                                   f"RCCL all-reduce + SGD), batch {batch} per GPU, synthetic data",
                       "global_batch": world * batch, "input_size": size, "parallelism": f"dp{world}",
                       "ranks_seen_by_collective_backend": seen,
                       "backend": torch.distributed.get_backend() if world > 1 else None,
                       "grad_payload": rec["grad_payload"], "gradient_buckets": rec["gradient_buckets"],
                       "loss_total": rec["loss_total"], "skipped_steps": rec["skipped_steps"]},
            "roofline": {"bound": "mfma", "achieved": rec["tflops_per_gpu"], "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": rec["frac"],
                         "note": "whole-step algorithmic conv flops / step time (not a single kernel)", "traffic": None}}))
    if comm is not None:
        torch.cuda.synchronize()
        Context.get(dev).lib.od_comm_destroy(comm)
    if world > 1:
        torch.distributed.destroy_process_group()


def run_infer(params, size, batch, inflight, steps, warmup, reps, rank, world, dev, graph=False, precision=None):
    """One step = one predict pass over a uint8 batch resident in HBM (network + decode + top-K + NMS).  With inflight > 1
    the step is queued on the next pipeline's stream and overlaps the tail of the previous steps; every step of a
    repetition is complete before its closing synchronize."""
    from object_detector_amd.detector import ObjectDetector
    od = ObjectDetector(params, batch, (size, size), device=dev, use_multi_gpu=world > 1, n_inflight=inflight,
                        precision=precision)
    rng = np.random.default_rng(1000 + rank)  # each rank its own shard of synthetic images
    x = torch.from_numpy(rng.integers(0, 256, (batch, size, size, 3), dtype=np.uint8)).to(dev)
    if od.n_inflight > 1:
        step = lambda: od.submit(x, conf_threshold=0.01, graph=graph)  # noqa: E731
    else:
        step = lambda: od.predict_batch_device(x, conf_threshold=0.01, graph=graph)  # noqa: E731
    times = timed_reps(step, steps, warmup, reps, world, dev)
    return od, times


def extra_block(params, a, dev):
    """The rest of BASELINE.json's metric ("images/sec at 320x320 & 640x640 batch-1 and batch-32" + the training configs),
    driver-timed in the same run: each entry is timed like the main line (warm-up, `reps` repetitions of K steps between
    synchronizes, median) on ONE GPU; `frac` = whole-network algorithmic conv flops / step time / 2.5 PFLOP/s."""
    out = {}
    net_flops_320 = 29.01e9 + 7.2e9

    def infer(key, size, batch, inflight, steps, precision=None):
        od, times = run_infer(params, size, batch, inflight, steps, max(3, steps // 5), a.reps, 0, 1, dev, precision=precision)
        el = _median(times)
        fl = net_flops_320 * (size / 320.0) ** 2 * batch
        what = "" if precision is None else (f", precision={precision} (every logit within 1e-3 x scale of the fp32 oracle: f32 "
                                             f"residual stream of stages 4-5, f32 FPN sums, split operands in 6 neck/head layers)")
        out[key] = {"workload": f"inference {size}x{size} batch {batch}, {inflight} in flight{what}", "images_per_sec": round(batch * steps / el, 1),
                    "ms_per_step": round(el / steps * 1e3, 4), "frac": round(fl * steps / el / 1e12 / PEAK_F16_TFLOPS, 4)}
        del od
        torch.cuda.empty_cache()

    infer("infer_320_b32_inflight1", 320, 32, 1, 20)
    infer("infer_320_b32_inflight3_mixed_precision", 320, 32, 3, 20, precision="mixed")
    infer("infer_640_b16_inflight3", 640, 16, 3, 12)
    infer("infer_640_b16_inflight3_mixed_precision", 640, 16, 3, 12, precision="mixed")
    infer("infer_640_b16_inflight1", 640, 16, 1, 12)
    infer("infer_320_b1_inflight3", 320, 1, 3, 100)
    infer("infer_320_b1_inflight1", 320, 1, 1, 100)
    infer("infer_640_b1_inflight3", 640, 1, 3, 60)
    infer("infer_640_b1_inflight1", 640, 1, 1, 60)
    for key, size, batch, steps in (("train_320_b32", 320, 32, 6), ("train_640_b16", 640, 16, 4)):
        rec, tr = run_train(params, size, batch, steps, 2, min(a.reps, 3), 0, 1, dev)
        out[key] = {"workload": f"training step {size}x{size} batch {batch} (assign + fwd + loss + bwd + SGD), 1 GPU",
                    "images_per_sec": rec["value"], "ms_per_step": rec["ms_per_step"], "frac": rec["frac"]}
        del tr
        torch.cuda.empty_cache()
    return out


XGMI_MESH_GBS = 7 * 153.0  # per GPU: 7 point-to-point links x ~153 GB/s (guides/MI355X_MICROARCH.md)


def multi_rank_train_block(params, a, rank, world, dev, backend):
    """N > 1 only: the path's ONE collective, timed on all N ranks inside the driver's default command (the headline stays
    inference, which has no data-path collective).  Training step of BASELINE.json configs[3] (32 x 320^2 per GPU) with the
    f32 and the bf16 gradient payload, bucketed-and-overlapped vs one collective after backward, and the all-reduce alone on
    the flat gradient buffer (173 MB f32 / 87 MB bf16) with its bus bandwidth against the 7-link xGMI mesh."""
    import ctypes as C
    from object_detector_amd import _lib
    from object_detector_amd.net import Context
    from object_detector_amd.trainer import init_comm
    ctx = Context.get(dev)
    comm, seen = None, torch.distributed.get_world_size()
    if backend == "nccl":
        comm, _ = init_comm(ctx)  # RCCL communicator through the C ABI (od_comm_*); the id travels over the process group
        r_, n_ = C.c_int(), C.c_int()
        _lib.check(ctx.lib.od_comm_count(comm, C.byref(r_), C.byref(n_)), "od_comm_count")
        seen = n_.value
        assert seen == world and r_.value == rank, f"RCCL reports rank {r_.value} of {seen}, launcher said {rank} of {world}"
    size, batch = 320, 32
    out = {"workload": f"Darknet53 {size}x{size} training step, batch {batch} per GPU (global {world * batch}), "
                       f"data-parallel gradient all-reduce", "collective": "od_allreduce (RCCL)" if comm is not None else
           f"torch.distributed {backend} (rehearsal: ranks share a GPU)", "od_comm_count": seen}
    saved = os.environ.get("OD_TRAIN_BUCKET_MB")
    try:
        for payload in ("f32", "bf16"):
            for bucket_mb, tag in ((32, "bucketed"), (0, "single")):
                os.environ["OD_TRAIN_BUCKET_MB"] = str(bucket_mb)
                rec, tr = run_train(params, size, batch, 6, 2, 3, rank, world, dev, comm, payload)
                key = f"step_{payload}_{tag}"
                out[key] = {"images_per_sec": rec["value"], "ms_per_step": rec["ms_per_step"],
                            "ms_per_step_reps": rec["ms_per_step_reps"], "collectives_per_step": max(1, rec["gradient_buckets"]),
                            "skipped_steps": rec["skipped_steps"]}
                if tag == "single":  # the all-reduce alone, on the same buffer and stream, 10 in a row between events
                    _barrier(world)
                    for _ in range(2):
                        tr._reduce_range(0, tr.n_flat)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    _barrier(world)
                    e0.record()
                    for _ in range(10):
                        tr._reduce_range(0, tr.n_flat)
                    e1.record()
                    torch.cuda.synchronize()
                    t = torch.tensor([e0.elapsed_time(e1) / 10.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    ms = float(t.item())
                    nbytes = tr.n_flat * (4 if payload == "f32" else 2)
                    busbw = 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9
                    out[f"allreduce_only_{payload}"] = {
                        "bytes": nbytes, "ms": round(ms, 4), "algbw_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                        "busbw_GBps": round(busbw, 1), "xgmi_mesh_GBps": XGMI_MESH_GBS, "frac_of_mesh": round(busbw / XGMI_MESH_GBS, 4),
                        "includes_bf16_casts": payload == "bf16"}
                del tr
                torch.cuda.empty_cache()
    finally:
        if saved is None:
            os.environ.pop("OD_TRAIN_BUCKET_MB", None)
        else:
            os.environ["OD_TRAIN_BUCKET_MB"] = saved
    if comm is not None:
        torch.cuda.synchronize()
        ctx.lib.od_comm_destroy(comm)
    return out


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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL: before the HIP runtime comes up
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))  # this process only waits for its N rank processes
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device(f"cuda:{local_rank % ndev}")  # one rank per GPU (the modulo only matters for 1-GPU rehearsals)
    backend = os.environ.get("OD_BENCH_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; "gloo" to rehearse on one GPU
    if world > 1:
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
    torch.cuda.set_device(dev)

    if a.mode == "train":
        return train_bench(a, rank, world, dev)

    from object_detector_amd import weights as W
    size = a.size
    batch = a.batch or (32 if size <= 320 else 16)
    params = W.random_init(2)
    od, times = run_infer(params, size, batch, a.inflight, a.steps, a.warmup, a.reps, rank, world, dev, a.graph)
    elapsed = _median(times)
    ranks_seen = world
    if world > 1:
        # no collective on the inference data path; what the process group itself reports, summed over the ranks
        t = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t)
        ranks_seen = int(round(float(t.item())))
        assert ranks_seen == torch.distributed.get_world_size() == world
    keep_count = od.post.keep_count.cpu().numpy()
    n_inflight, precision, ablated = od.n_inflight, od.precision, od.net.ablated

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
        prof = pathlib.Path(__file__).resolve().parent / "profiles"
        pj = next((q for q in (prof / "r03" / "pmc_traffic.json", prof / "r02" / "pmc_traffic.json",
                                   prof / "r01" / "pmc_traffic.json") if q.exists()), None)
        if size == 320 and batch == 32 and pj is not None:
            for kname, rec in json.loads(pj.read_text())["kernels"].items():
                if dom in kname:
                    traffic = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"],
                               "algorithmic_bytes_per_launch": round(g["bytes"] / g["n"]),
                               "source": str(pj.relative_to(prof.parent))}
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

    out = None
    if rank == 0:
        total_images = world * batch * a.steps
        out = {
            "metric": "images_per_sec",
            "value": round(total_images / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "reps": a.reps,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "ms_per_step_reps": [round(t / a.steps * 1e3, 4) for t in times],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"Darknet53 {size}x{size} inference (backbone + neck + shared head + decode + "
                                   f"top-k + NMS), batch {batch} per GPU, synthetic VOC-shaped uint8 input, "
                                   f"random-init weights",
                       "global_batch": world * batch, "input_size": size, "parallelism": f"dp{world}",
                       "graph": bool(a.graph), "batches_in_flight": n_inflight,
                       "ranks_seen_by_process_group": ranks_seen, "backend": backend if world > 1 else None,
                       "kept_boxes_rank0_img0": int(keep_count[0]), "precision": precision,
                       **({"ABLATED_TIMING_ONLY": os.environ.get("OD_ABLATE_OPS")} if ablated else {})},
            "roofline": {"bound": "mfma", "kernel": dom, "launches_per_step": g["n"],
                         "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16_TFLOPS, 4),
                         "avg_launch_us": round(g["ms"] / g["n"] * 1e3, 2),
                         "cu_fill": round(cu_fill, 3), "achieved_on_occupied_cus": round(achieved_active, 2),
                         "algorithmic_gflop_per_launch": round(g["flops"] / g["n"] / 1e9, 3),
                         "network_ms_per_batch": round(net_ms, 4),
                         "network_tflops": round(sum(i["flops"] for i in info) / (net_ms * 1e-3) / 1e12, 2),
                         "traffic": traffic,
                         # all instantiations of the 8-wave kernel together (one source; the symbols differ in tile height,
                         # fused second layer, grouped launch): what the kernel as such sustains over its launches of a step
                         "kernel_family": (lambda fam: {"kernel": dom.split("<")[0], "launches_per_step": sum(v["n"] for v in fam),
                                                        "achieved": round(sum(v["flops"] for v in fam) / (sum(v["ms"] for v in fam) * 1e-3) / 1e12, 2),
                                                        "frac": round(sum(v["flops"] for v in fam) / (sum(v["ms"] for v in fam) * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4)})(
                             [v for k, v in groups.items() if k.split("<")[0] == dom.split("<")[0]]),
                         # the other instantiations of the same 8-wave kernel (since round 2 the stage-3 launches also run
                         # the consuming 1x1 layer in their epilogue and are a kernel symbol of their own)
                         "same_kernel_other_instantiations": [
                             {"kernel": k, "launches_per_step": v["n"], "avg_launch_us": round(v["ms"] / v["n"] * 1e3, 2),
                              "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                              "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4)}
                             for k, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
                             if k != dom and k.split("<")[0] == dom.split("<")[0]][:3]},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(size, a.cpu_seconds)
        if world == 1 and not a.no_extra:
            del od
            torch.cuda.empty_cache()
            out["extra"] = extra_block(params, a, dev)
    if world > 1 and not a.no_extra:
        # N > 1: the training step's gradient all-reduce, timed on ALL ranks (every rank takes part in the collectives; rank
        # 0 prints).  The headline above is complete at this point, and this block has never run on more than one GPU before
        # the driver does: a watchdog makes sure the ONE line still goes out (with the error noted) if a rank fails or a
        # collective hangs, instead of the whole multi-GPU measurement being lost with it.
        import threading
        limit = float(os.environ.get("OD_BENCH_TRAIN_TIMEOUT", "300"))
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                out["train"] = {"error": f"the multi-rank train block did not finish within {limit:.0f} s (a rank failed or a "
                                         f"collective hung); the inference headline above is unaffected"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        timer = threading.Timer(limit, bail)
        timer.daemon = True
        timer.start()
        try:
            tb = multi_rank_train_block(params, a, rank, world, dev, backend)
        except Exception as e:  # noqa: BLE001 -- whatever it is, the headline line must still be printed
            tb = {"error": f"{type(e).__name__}: {e}"}
            if rank != 0:  # the other ranks may be waiting in a collective for this one: their watchdogs end them
                finished.set()
                timer.cancel()
                os._exit(0)
        finished.set()
        timer.cancel()
        if rank == 0:
            out["train"] = tb
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        if out is not None and isinstance(out.get("train"), dict) and "error" in out["train"]:
            os._exit(0)  # do not wait in destroy_process_group for ranks that are gone
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
