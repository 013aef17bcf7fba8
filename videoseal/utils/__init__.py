from . import cfg  # noqa: F401

from .._overlay import extend as _extend, fallback_getattr as _fallback  # noqa: E402
_extend(__path__, "utils")
__getattr__ = _fallback(__name__, "utils")
