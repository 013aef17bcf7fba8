from . import metrics  # noqa: F401

from .._overlay import extend as _extend, fallback_getattr as _fallback_pkg  # noqa: E402
_extend(__path__, "evals")
__getattr__ = _fallback_pkg(__name__, "evals")
