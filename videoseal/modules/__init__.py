from . import jnd  # noqa: F401
