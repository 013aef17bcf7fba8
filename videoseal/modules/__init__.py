from . import jnd  # noqa: F401

from .._overlay import extend as _extend  # noqa: E402
_extend(__path__, "modules")
