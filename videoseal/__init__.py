"""``import videoseal`` compatibility package: the reference's import paths bound to the MI355X implementation.

Callers written against facebookresearch/videoseal keep their imports (inference_streaming.py:18-20, inference_av.py:20-21,
README quick start):

    import videoseal
    from videoseal.models import Videoseal
    from videoseal.evals.metrics import bit_accuracy, psnr
    from videoseal.utils.cfg import setup_model_from_model_card
    from videoseal.augmentation import get_validation_augs
    model = videoseal.load("videoseal")          # videoseal/__init__.py:13-17

Everything here is a re-export of ``videoseal_amd`` (host plumbing over libvideoseal_hip.so); there is no second code path.
Modules outside the path (``videoseal.losses``, ``videoseal.data``, ``videoseal.utils.optim`` ...: what train.py:55-72 imports besides
the model) resolve to a reference checkout named by ``VIDEOSEAL_REFERENCE_ROOT`` when that is set (videoseal/_overlay.py).
"""
from videoseal_amd import __version__, available_cards, build, load  # noqa: F401

from . import augmentation, evals, models, modules, utils  # noqa: E402,F401

from ._overlay import extend as _extend, fallback_getattr as _fallback  # noqa: E402
_extend(__path__)
__getattr__ = _fallback(__name__, "")
