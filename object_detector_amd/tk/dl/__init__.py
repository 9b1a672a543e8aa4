"""tk.dl: session() scope (reference voc_validate.py:17) and the `od` sub-namespace."""
import contextlib

from . import od  # noqa: F401


@contextlib.contextmanager
def session(device=None):
    """The reference opens a TF session here; on MI355X it pins the process to its GPU and drains it on exit."""
    import os

    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) if device is None else device)
    try:
        yield
    finally:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
