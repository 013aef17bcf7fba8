"""videoseal.augmentation.augmenter (augmentation/augmenter.py:26-199)."""
from videoseal_amd.augmentation import Augmenter, get_dummy_augmenter, name2aug, video_augs  # noqa: F401

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "augmentation/augmenter.py")
