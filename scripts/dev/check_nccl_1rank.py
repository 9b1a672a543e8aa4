#!/usr/bin/env python3
"""One-rank sanity check of the collective backend bench.py uses under a multi-rank launch: torch.distributed 'nccl' (= RCCL)
bound to cuda:0, barrier + MAX all-reduce of a device tensor -- the exact calls of bench.py's timed_reps / main."""
import os
import torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0")
torch.distributed.init_process_group("nccl", device_id=dev)
torch.cuda.set_device(dev)
torch.distributed.barrier()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
u = torch.ones(1, dtype=torch.float64, device=dev)
torch.distributed.all_reduce(u)
print("nccl 1-rank ok:", t.tolist(), u.item(), torch.distributed.get_backend())
torch.distributed.destroy_process_group()
