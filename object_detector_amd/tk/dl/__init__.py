"""tk.dl: session() scope (reference voc_validate.py:17) and the `od` sub-namespace."""
import contextlib
import os

from . import od  # noqa: F401


def dist_env():
    """(rank, local_rank, world) as the launcher (torch.distributed.run / torchrun) exported them; (0, 0, 1) when alone."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def is_main_process():
    """True on the rank that writes files (logs, result images): rank 0, or a process that is not part of a job."""
    import torch
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank() == 0
    return dist_env()[0] == 0


@contextlib.contextmanager
def session(device=None):
    """The reference opens a TF session here (and `use_multi_gpu=True` replicates the model inside it,
    voc_validate.py:17,26).  On MI355X the unit is one process per GPU: when the launcher started several ranks
    (WORLD_SIZE > 1) and no process group exists yet, this scope creates it BEFORE the first GPU call -- backend "nccl"
    (= RCCL over xGMI) bound to this rank's GPU, or $OD_DIST_BACKEND / $OD_BENCH_BACKEND (e.g. "gloo" to rehearse several
    ranks on one GPU or on CPUs) -- pins the process to its GPU, and on exit drains the GPU and destroys the group it
    created.  `ObjectDetector(use_multi_gpu=True).predict` then shards the images over the ranks (detector.dist_info)."""
    import torch
    rank, local_rank, world = dist_env()
    created = False
    ndev = torch.cuda.device_count()  # only counts devices; this process has made no other GPU call yet
    # ONE device decision for both the process group and set_device: the caller's, else this rank's GPU
    dev = None
    if ndev > 0:
        dev = torch.device(device) if device is not None else torch.device(f"cuda:{local_rank % ndev}")
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device(f"cuda:{local_rank % ndev}")
    if world > 1 and torch.distributed.is_available() and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC; must be set before the runtime comes up
        backend = os.environ.get("OD_DIST_BACKEND") or os.environ.get("OD_BENCH_BACKEND") or ("nccl" if ndev > 0 else "gloo")
        if backend == "nccl":
            if dev is None or dev.type != "cuda":
                from ..._lib import OdError
                raise OdError("backend nccl (RCCL) was requested but this process sees no GPU; set OD_DIST_BACKEND=gloo "
                              "for a CPU rehearsal")
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        created = True
    if dev is not None and dev.type == "cuda" and torch.cuda.is_available():
        torch.cuda.set_device(dev)
    try:
        yield
    finally:
        if ndev > 0 and torch.cuda.is_available():
            torch.cuda.synchronize()
        if created:
            torch.distributed.destroy_process_group()
