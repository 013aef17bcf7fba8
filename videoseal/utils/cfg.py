"""videoseal.utils.cfg (utils/cfg.py:181-251): card name / Path -> Videoseal with the checkpoint loaded (strict=False)."""
from videoseal_amd import load as setup_model_from_model_card  # noqa: F401
