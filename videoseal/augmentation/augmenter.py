"""videoseal.augmentation.augmenter (augmentation/augmenter.py:26-199)."""
from videoseal_amd.augmentation import Augmenter, get_dummy_augmenter, name2aug, video_augs  # noqa: F401
