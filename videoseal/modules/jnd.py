"""videoseal.modules.jnd (modules/jnd.py:11-114)."""
from videoseal_amd.model import JND  # noqa: F401
