"""videoseal.models.embedder (models/embedder.py:130-282): the U-Net embedder and its factory on the HIP path."""
from videoseal_amd.builders import build_embedder  # noqa: F401
from videoseal_amd.model import Embedder  # noqa: F401
UnetEmbedder = Embedder

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "models/embedder.py")
