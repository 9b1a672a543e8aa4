"""pytoolkit-shaped namespace: exactly the `tk.*` symbols the reference scripts call:
tk.dl.od.{ObjectDetector, od_gen}, tk.dl.session, tk.data.voc.{load_07_test, evaluate, CLASS_NAMES},
tk.ml.{compute_scores, print_scores, plot_objects}, tk.ndimage.save, tk.log.{init, get, trace}, tk.tqdm,
tk.better_exceptions.  `import pytoolkit as tk` resolves here through the alias package at the repo root."""
from . import data, dl, log, ml, ndimage  # noqa: F401


def better_exceptions():
    """reference voc_validate.py:10 -- pretty tracebacks; nothing to do here."""


def tqdm(iterable=None, **kw):
    """reference check_assign.py:23"""
    from tqdm import tqdm as _tqdm
    return _tqdm(iterable, **kw)
