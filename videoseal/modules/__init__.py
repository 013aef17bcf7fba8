from . import jnd  # noqa: F401

from .._overlay import extend as _extend, fallback_getattr as _fallback_pkg  # noqa: E402
_extend(__path__, "modules")
__getattr__ = _fallback_pkg(__name__, "modules")
