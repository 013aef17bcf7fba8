from . import cfg  # noqa: F401
