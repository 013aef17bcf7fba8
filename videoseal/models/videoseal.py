"""videoseal.models.videoseal (models/videoseal.py:15-428)."""
from videoseal_amd.model import Videoseal  # noqa: F401

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "models/videoseal.py")
