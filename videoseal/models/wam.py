"""videoseal.models.wam (models/wam.py:18-234)."""
from videoseal_amd.model import Wam  # noqa: F401
