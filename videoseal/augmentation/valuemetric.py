"""videoseal.augmentation.valuemetric (augmentation/valuemetric.py of the reference): the same class names on the HIP kernels."""
from videoseal_amd.augmentation import *  # noqa: F401,F403
