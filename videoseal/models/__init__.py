"""videoseal.models (models/__init__.py of the reference): the module classes of the embed / extract path."""
from videoseal_amd.model import Blender, Embedder, Extractor, Videoseal, Wam, build_model  # noqa: F401
from .videoseal import Videoseal as _V  # noqa: F401  (videoseal.models.videoseal.Videoseal resolves too)
