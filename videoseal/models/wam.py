"""videoseal.models.wam (models/wam.py:18-234)."""
from videoseal_amd.model import Wam  # noqa: F401

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "models/wam.py")
