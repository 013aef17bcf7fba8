"""videoseal.utils.cfg (utils/cfg.py:28-251): card / checkpoint -> Videoseal with the weights loaded (strict=False), on the HIP path.
`setup_dataset` and the download helpers (datasets, network: outside the path) resolve to the checkout's file under VIDEOSEAL_REFERENCE_ROOT."""
from videoseal_amd.cfg import (DEFAULT_CARD, SubModelConfig, VideosealConfig, get_config_from_checkpoint, resolve_config_path,  # noqa: F401
                               setup_model, setup_model_from_checkpoint, setup_model_from_model_card)

from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "utils/cfg.py")
