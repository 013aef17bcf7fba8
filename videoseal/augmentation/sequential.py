"""videoseal.augmentation.sequential (augmentation/sequential.py:8-30)."""
from videoseal_amd.augmentation import Sequential  # noqa: F401
