"""videoseal.models (models/__init__.py of the reference): the module classes of the embed / extract path."""
from videoseal_amd.builders import build_embedder, build_extractor  # noqa: F401  (train.py:67, 266, 281)
from videoseal_amd.model import Blender, Embedder, Extractor, Videoseal, Wam, build_model  # noqa: F401
from .videoseal import Videoseal as _V  # noqa: F401  (videoseal.models.videoseal.Videoseal resolves too)

from .._overlay import extend as _extend, fallback_getattr as _fallback_pkg  # noqa: E402
_extend(__path__, "models")
__getattr__ = _fallback_pkg(__name__, "models")
