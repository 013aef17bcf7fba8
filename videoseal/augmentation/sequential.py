"""videoseal.augmentation.sequential (augmentation/sequential.py:8-30)."""
from videoseal_amd.augmentation import Sequential  # noqa: F401

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "augmentation/sequential.py")
