"""videoseal.evals.metrics (evals/metrics.py): the metrics train.py:65 / evals/full.py:46 import, on plain torch (videoseal_amd/metrics.py).
Names outside that set (`vmaf_on_tensor`, `bd_rate`, `bit_accuracy_inference`, ...: ffmpeg / scipy tooling) resolve to the checkout's own
file when VIDEOSEAL_REFERENCE_ROOT is set; its `import pytorch_msssim` is served by this package's restatement when that package is absent."""
from videoseal_amd.metrics import (accuracy, bit_accuracy, bit_accuracy_1msg, capacity, iou, linf, msssim, plogp, psnr,  # noqa: F401
                                   pvalue, ssim)


def _msssim_stub():
    import types
    from videoseal_amd import metrics as m
    mod = types.ModuleType("pytorch_msssim")
    mod.__doc__ = "stand-in installed by the videoseal shim (videoseal_amd.metrics restates the algorithm); the real package is absent"

    def _avg(v, size_average):
        return v.mean() if size_average else v
    mod.ssim = lambda X, Y, data_range=255, size_average=True, **kw: _avg(m.ssim(X, Y, data_range), size_average)
    mod.ms_ssim = lambda X, Y, data_range=255, size_average=True, **kw: _avg(m.msssim(X, Y, data_range), size_average)
    return mod


from .._overlay import fallback_module_getattr as _fallback  # noqa: E402
__getattr__ = _fallback(__name__, "evals/metrics.py", stubs={"pytorch_msssim": _msssim_stub})
