"""tk.log: file+console logger and a @trace decorator (reference voc_validate.py:18,22,30)."""
import functools
import logging
import pathlib
import time

_ROOT = "od"


def init(path=None, level=logging.INFO):
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
