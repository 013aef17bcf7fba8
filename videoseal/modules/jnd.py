"""videoseal.modules.jnd (modules/jnd.py:11-114)."""
from videoseal_amd.model import JND  # noqa: F401

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "modules/jnd.py")
