"""videoseal.models.extractor (models/extractor.py:40-213): the ConvNeXt-V2 / ViT extractors and their factory on the HIP path."""
from videoseal_amd.builders import build_extractor  # noqa: F401
from videoseal_amd.model import Extractor  # noqa: F401
ConvnextExtractor = SegmentationExtractor = Extractor

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "models/extractor.py")
