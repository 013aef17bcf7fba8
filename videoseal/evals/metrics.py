"""videoseal.evals.metrics (evals/metrics.py:22-36, 150-178): the two parity metrics of the path."""
from videoseal_amd.metrics import bit_accuracy, psnr  # noqa: F401
