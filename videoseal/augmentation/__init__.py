"""videoseal.augmentation (augmentation/__init__.py, augmenter.py, sequential.py, valuemetric.py, geometric.py, video.py)."""
from videoseal_amd.augmentation import *  # noqa: F401,F403
from videoseal_amd.augmentation import (Augmenter, Sequential, get_dummy_augmenter, get_validation_augs, name2aug)  # noqa: F401
from . import augmenter, sequential  # noqa: E402,F401

from .._overlay import extend as _extend, fallback_getattr as _fallback_pkg  # noqa: E402
_extend(__path__, "augmentation")
__getattr__ = _fallback_pkg(__name__, "augmentation")
