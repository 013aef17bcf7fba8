"""videoseal.augmentation.video (augmentation/video.py of the reference): the same class names on the HIP kernels."""
from videoseal_amd.augmentation import *  # noqa: F401,F403

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "augmentation/video.py")
