"""videoseal.models.videoseal (models/videoseal.py:15-428)."""
from videoseal_amd.model import Videoseal  # noqa: F401
