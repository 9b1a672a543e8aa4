"""Alias so that the reference's `import pytoolkit as tk` (voc_validate.py:6) resolves to this build's
pytoolkit-shaped namespace when the repository root is on sys.path.  (The reference's own pytoolkit is an
un-vendored, empty submodule: SURVEY.md §0.)"""
import sys as _sys

from object_detector_amd import tk as _tk
from object_detector_amd.tk import *  # noqa: F401,F403
from object_detector_amd.tk import better_exceptions, data, dl, log, ml, ndimage, tqdm  # noqa: F401

for _n in ("data", "dl", "log", "ml", "ndimage"):
    _sys.modules[f"pytoolkit.{_n}"] = getattr(_tk, _n)
_sys.modules["pytoolkit.data.voc"] = _tk.data.voc
_sys.modules["pytoolkit.dl.od"] = _tk.dl.od
