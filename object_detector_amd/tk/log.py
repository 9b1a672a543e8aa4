"""tk.log: file+console logger and a @trace decorator (reference voc_validate.py:18,22,30)."""
import functools
import logging
import pathlib
import time

_ROOT = "od"


def init(path=None, level=logging.INFO):
    """Console handler on every rank; the FILE only on the main process (rank 0): in a multi-rank job every rank runs the
    same script, and N ranks opening the same log with mode "w" would truncate each other."""
    from .dl import is_main_process
    if not is_main_process():
        path = None
    logger = logging.getLogger(_ROOT)
    logger.setLevel(level)
    logger.handlers.clear()
    fmt = logging.Formatter("[%(asctime)s %(levelname)s] %(message)s")
    sh = logging.StreamHandler()
    sh.setFormatter(fmt)
    logger.addHandler(sh)
    if path is not None:
        path = pathlib.Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        fh = logging.FileHandler(path, mode="w", encoding="utf-8")
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger


def get(name=None):
    return logging.getLogger(_ROOT if not name else f"{_ROOT}.{name}")


def trace(process_name=None):
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **kw):
            lg = get(fn.__module__)
            nm = process_name or fn.__qualname__
            lg.debug("%s start", nm)
            t = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                lg.info("%s done in %.3f s", nm, time.perf_counter() - t)
        return wrapper
    return deco
