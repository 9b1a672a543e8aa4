from . import voc  # noqa: F401
