"""Augmentations of the embed -> augment -> extract path on the HIP kernels (csrc/aug.hip).

Same call surface as the reference's ``videoseal.augmentation`` for the rows in scope:
every op is ``op(image, mask=None, strength=None) -> (image, mask)``, ``Sequential(*ops)(image, mask, args)``,
``Augmenter(masks, augs, augs_params, num_augs)`` with the multinomial pick, and the validation tables.
Random parameters are drawn exactly like the reference (torch CPU RNG: ``torch.randint`` / ``torch.rand``),
so a seeded run picks the same strengths.  Out of scope and loud: the H.264/H.265/VP9/AV1 codecs (external libx264,
SURVEY 8(f)2).  Rotate / Perspective follow torchvision's grid construction + ATen grid_sample (torchvision itself is
not vendored by the reference: pinned against oracle/augment.py only).

Inside a differentiable forward (model.forward under autograd, videoseal_amd/autograd.py) every op is a graph node whose backward is a HIP
kernel too: Crop / HorizontalFlip / Resize / Brightness / Contrast / Saturation / Grayscale / Hue / GaussianBlur / Rotate / Perspective and the
temporal ops (DropFrame / SpeedChange / TemporalReorder / WindowAveraging) have exact adjoints (csrc/bwd_shell.hip), JPEG / MedianFilter /
GaussianNoise / the codecs are the reference's straight-through estimators (forward value = codec / filter output, identity gradient); only
`passthrough=False` JPEG / MedianFilter -- which the reference cannot differentiate either -- raise in backward instead of cutting the graph.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as AG
from . import native as N

FUSE = os.environ.get("VIDEOSEAL_AUG_FUSE", "1") != "0"     # Sequential: consecutive Crop / Resize / colour ops in fused passes (round 5)
TIMERS = None         # bench.py: a list -> every fused / JPEG launch group appends (name, start event, end event, algorithmic bytes)


COLOR_OPS = {"brightness": 0, "contrast": 1, "saturation": 2, "hue": 3, "grayscale": 4}


def _dev(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise N.NativeError("augmentations run on the HIP kernels: pass device tensors (no CPU fallback)")
    return N.f32c(x)


def _planes(x: torch.Tensor) -> Tuple[int, int, int]:
    return x.shape[0] * x.shape[1], x.shape[-2], x.shape[-1]


def color_op(x: torch.Tensor, op: str, factor: float) -> torch.Tensor:
    if AG.needs_grad(x):
        return AG.ColorFn.apply(x, op, factor)
    x = _dev(x)
    F_, Cc, H, W = x.shape
    if Cc != 3:
        raise ValueError("colour augmentations expect 3-channel frames")
    L = N.lib()
    out = torch.empty_like(x)
    scratch = torch.empty(int(L.vs_aug_color_scratch_floats(F_, H, W)), device=x.device, dtype=torch.float32)
    N.check(L.vs_aug_color(N.ptr(x), N.ptr(out), F_, H, W, COLOR_OPS[op], float(factor), N.ptr(scratch), N.stream()), "vs_aug_color")
    return out


def crop_flip(x: torch.Tensor, i: int, j: int, h: int, w: int, flip: bool = False) -> torch.Tensor:
    if AG.needs_grad(x):
        return AG.CropFlipFn.apply(x, i, j, h, w, flip)
    x = _dev(x)
    planes, H, W = _planes(x)
    out = torch.empty(x.shape[0], x.shape[1], h, w, device=x.device, dtype=torch.float32)
    N.check(N.lib().vs_aug_crop_flip(N.ptr(x), N.ptr(out), planes, H, W, i, j, h, w, int(flip), N.stream()), "vs_aug_crop_flip")
    return out


def resize(x: torch.Tensor, size: Tuple[int, int], antialias: bool = True) -> torch.Tensor:
    if AG.needs_grad(x):
        return AG.ResizeFn.apply(x, tuple(size), antialias)
    x = _dev(x)
    planes, H, W = _planes(x)
    out = torch.empty(x.shape[0], x.shape[1], size[0], size[1], device=x.device, dtype=torch.float32)
    if FUSE and x.dim() == 4 and x.shape[1] == 3:      # 3-channel frames: the LDS-tiled kernel (tap weights once per tile; bit-identical values)
        rc = N.lib().vs_aug_crop_resize_color(N.ptr(x), N.ptr(out), x.shape[0], H, W, 0, 0, H, W, size[0], size[1], int(antialias), 0, None, None, N.stream())
        if rc != N.ERR_UNSUPPORTED:
            N.check(rc, "vs_aug_crop_resize_color")
            return out
    N.check(N.lib().vs_resize_nchw(N.ptr(x), N.ptr(out), planes, H, W, size[0], size[1], int(antialias), N.stream()), "vs_resize_nchw")
    return out


def _timed(name: str, nbytes: int, fn):
    if TIMERS is None or torch.cuda.is_current_stream_capturing():
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    TIMERS.append((name, e0, e1, nbytes))
    return out


def color_chain(x: torch.Tensor, ops: List[Tuple[str, float]]) -> torch.Tensor:
    """the colour ops `ops` = [(name, factor), ...] applied in order.  Runs of ops are ONE pass over the clip each (vs_aug_color_chain); a run ends
    in front of every 'contrast' (it needs the mean of its own input).  Same expressions in the same order as color_op -> bit-identical."""
    if not ops:
        return x
    if AG.needs_grad(x) or not FUSE or len(ops) == 1:
        for name, f in ops:
            x = color_op(x, name, f)
        return x
    x = _dev(x)
    F_, Cc, H, W = x.shape
    if Cc != 3:
        raise ValueError("colour augmentations expect 3-channel frames")
    L = N.lib()
    runs: List[List[Tuple[str, float]]] = []
    for name, f in ops:
        if not runs or name == "contrast" or len(runs[-1]) == 6:
            runs.append([])
        runs[-1].append((name, float(f)))
    for run in runs:
        out = torch.empty_like(x)
        scratch = torch.empty(int(L.vs_aug_color_scratch_floats(F_, H, W)), device=x.device, dtype=torch.float32) if run[0][0] == "contrast" else None
        oa = (C.c_int * len(run))(*[COLOR_OPS[n] for n, _ in run])
        fa = (C.c_float * len(run))(*[f for _, f in run])
        src = x
        x = _timed("aug:" + "+".join(n for n, _ in run), (3 if scratch is not None else 2) * x.numel() * 4,
                   lambda: (N.check(L.vs_aug_color_chain(N.ptr(src), N.ptr(out), F_, H, W, len(run), oa, fa, N.ptr(scratch), N.stream()),
                                    "vs_aug_color_chain"), out)[1])
    return x


def crop_resize_color(x: torch.Tensor, crop: Optional[Tuple[int, int, int, int]], size: Tuple[int, int], ops: List[Tuple[str, float]],
                      antialias: bool = True) -> torch.Tensor:
    """crop window (i, j, h, w) (None = the whole frame) -> resize to `size` -> colour ops, fused: one kernel for crop + resize + the colour ops
    in front of the first 'contrast' (vs_aug_crop_resize_color: the cropped clip is never materialised), color_chain for the rest.  Falls back
    to the separate launches where the fused kernel does not apply (gradients wanted, window outside the frame, extreme down-scaling)."""
    def separate():
        y = crop_flip(x, *crop) if crop is not None else x
        return color_chain(resize(y, size, antialias), ops)
    if AG.needs_grad(x) or not FUSE or x.shape[1] != 3:
        return separate()
    x = _dev(x)
    F_, _, H, W = x.shape
    i, j, h, w = crop if crop is not None else (0, 0, H, W)
    if i < 0 or j < 0 or i + h > H or j + w > W:
        return separate()
    k = 0
    while k < len(ops) and ops[k][0] != "contrast" and k < 6:
        k += 1
    head, tail = ops[:k], ops[k:]
    out = torch.empty(F_, 3, size[0], size[1], device=x.device, dtype=torch.float32)
    oa = (C.c_int * max(1, k))(*[COLOR_OPS[n] for n, _ in head])
    fa = (C.c_float * max(1, k))(*[float(f) for _, f in head])
    name = "aug:" + "+".join((["crop"] if crop is not None else []) + ["resize"] + [n for n, _ in head])
    rc = _timed(name, F_ * 3 * h * w * 4 + out.numel() * 4,
                lambda: N.lib().vs_aug_crop_resize_color(N.ptr(x), N.ptr(out), F_, H, W, i, j, h, w, size[0], size[1], int(antialias), k, oa, fa, N.stream()))
    if rc == N.ERR_UNSUPPORTED:
        if TIMERS is not None and TIMERS and TIMERS[-1][0] == name:
            TIMERS.pop()
        return separate()
    N.check(rc, "vs_aug_crop_resize_color")
    return color_chain(out, tail)


def _ste(x: torch.Tensor, fn, clamp01: bool = False) -> torch.Tensor:
    """`x + (fn(x) - x).detach()` of valuemetric.py:35, 90 / video.py:113: forward value fn(x), identity gradient (masked to 0 <= x <= 1 when
    the op clamps first)"""
    if AG.needs_grad(x):
        with torch.no_grad():
            y = fn(x)
        return AG.SteFn.apply(x, y, clamp01)
    return fn(x)


def _no_adjoint(fn, what: str):
    """forward values from the kernel; inside a differentiable forward the node raises in backward instead of cutting the graph"""
    def wrapped(x, *a, **k):
        if AG.needs_grad(x):
            with torch.no_grad():
                y = fn(x, *a, **k)
            return AG.NoAdjointFn.apply(x, y, what)
        return fn(x, *a, **k)
    wrapped.__doc__, wrapped.__name__ = fn.__doc__, fn.__name__
    return wrapped


def gaussian_blur(x: torch.Tensor, kernel_size: int) -> torch.Tensor:
    if AG.needs_grad(x):
        return AG.BlurFn.apply(x, kernel_size)
    x = _dev(x)
    planes, H, W = _planes(x)
    sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8            # torchvision default when sigma is None
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    N.check(N.lib().vs_gaussian_blur(N.ptr(x), N.ptr(tmp), N.ptr(out), planes, H, W, kernel_size, sigma, N.stream()), "vs_gaussian_blur")
    return out


def median_filter(x: torch.Tensor, kernel_size: int) -> torch.Tensor:
    if kernel_size % 2 == 0:
        raise ValueError("Kernel size must be odd.")
    x = _dev(x)
    planes, H, W = _planes(x)
    out = torch.empty_like(x)
    N.check(N.lib().vs_median_filter(N.ptr(x), N.ptr(out), planes, H, W, kernel_size, N.stream()), "vs_median_filter")
    return out


def jpeg_compress(x: torch.Tensor, quality: int) -> torch.Tensor:
    """[F,3,H,W] in [0,1] (values outside are clamped, valuemetric.py:41) -> libjpeg round trip at `quality`."""
    x = _dev(x)
    F_, Cc, H, W = x.shape
    if Cc != 3:
        raise ValueError("JPEG expects 3-channel frames")
    L = N.lib()
    ws = torch.empty(int(L.vs_jpeg_workspace_bytes(F_, H, W)), device=x.device, dtype=torch.uint8)
    out = torch.empty_like(x)
    N.check(L.vs_jpeg_roundtrip(N.ptr(x), N.ptr(out), F_, H, W, int(quality), N.ptr(ws), N.stream()), "vs_jpeg_roundtrip")
    return out


# ------------------------------------------------------------------------------------------------ op classes
class _Aug(nn.Module):
    def __repr__(self):
        return self.__class__.__name__


class Identity(_Aug):
    def forward(self, image, mask=None, *args, **kwargs):
        return image, mask


class HorizontalFlip(_Aug):
    def forward(self, image, mask=None, *args, **kwargs):
        H, W = image.shape[-2:]
        image = crop_flip(image, 0, 0, H, W, flip=True)
        mask = crop_flip(mask, 0, 0, H, W, flip=True) if mask is not None else mask
        return image, mask


class _Sized(_Aug):
    def __init__(self, min_size=None, max_size=None):
        super().__init__()
        self.min_size, self.max_size = min_size, max_size

    def get_random_size(self, h, w):
        if self.min_size is None or self.max_size is None:
            raise ValueError("min_size and max_size must be provided")
        return (torch.randint(int(self.min_size * h), int(self.max_size * h) + 1, size=(1,)).item(),
                torch.randint(int(self.min_size * w), int(self.max_size * w) + 1, size=(1,)).item())

    def _out_size(self, image, size):
        h, w = image.shape[-2:]
        return self.get_random_size(h, w) if size is None else (int(size * h), int(size * w))


class Resize(_Sized):
    def forward(self, image, mask=None, size=None):
        out = self._out_size(image, size)
        return resize(image, out, True), (resize(mask, out, True) if mask is not None else mask)

    def plan(self, shape, size=None):
        """the parameters forward() would draw for an input of `shape` (same random draws, same order): ('resize', (oh, ow))"""
        h, w = shape[-2:]
        return ("resize", self.get_random_size(h, w) if size is None else (int(size * h), int(size * w)))


class Crop(_Sized):
    def forward(self, image, mask=None, size=None):
        th, tw = self._out_size(image, size)
        h, w = image.shape[-2:]
        if h < th or w < tw:
            raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
        if w == tw and h == th:                              # torchvision RandomCrop.get_params
            i, j = 0, 0
        else:
            i = torch.randint(0, h - th + 1, size=(1,)).item()
            j = torch.randint(0, w - tw + 1, size=(1,)).item()
        return crop_flip(image, i, j, th, tw), (crop_flip(mask, i, j, th, tw) if mask is not None else mask)

    def plan(self, shape, size=None):
        """('crop', (i, j, th, tw)) with forward()'s draws"""
        h, w = shape[-2:]
        th, tw = self.get_random_size(h, w) if size is None else (int(size * h), int(size * w))
        if h < th or w < tw:
            raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
        if w == tw and h == th:
            return ("crop", (0, 0, th, tw))
        return ("crop", (torch.randint(0, h - th + 1, size=(1,)).item(), torch.randint(0, w - tw + 1, size=(1,)).item(), th, tw))


class _Factor(_Aug):
    op = ""

    def __init__(self, min_factor=None, max_factor=None):
        super().__init__()
        self.min_factor, self.max_factor = min_factor, max_factor

    def get_random_factor(self):
        if self.min_factor is None or self.max_factor is None:
            raise ValueError("min_factor and max_factor must be provided")
        return torch.rand(1).item() * (self.max_factor - self.min_factor) + self.min_factor

    def forward(self, image, mask=None, factor=None):
        factor = self.get_random_factor() if factor is None else factor
        return color_op(image, self.op, factor), mask

    def plan(self, shape, factor=None):
        return ("color", (self.op, self.get_random_factor() if factor is None else factor))


class Brightness(_Factor):
    op = "brightness"


class Contrast(_Factor):
    op = "contrast"


class Saturation(_Factor):
    op = "saturation"


class Hue(_Factor):
    op = "hue"


class Grayscale(_Aug):
    def forward(self, image, mask=None, *args, **kwargs):
        return color_op(image, "grayscale", 0.0), mask


class _Kernel(_Aug):
    def __init__(self, min_kernel_size=None, max_kernel_size=None, passthrough=True):
        super().__init__()
        self.min_kernel_size, self.max_kernel_size, self.passthrough = min_kernel_size, max_kernel_size, passthrough

    def get_random_kernel_size(self):
        if self.min_kernel_size is None or self.max_kernel_size is None:
            raise ValueError("Kernel size range must be specified")
        k = torch.randint(self.min_kernel_size, self.max_kernel_size + 1, size=(1,)).item()
        return k + 1 if k % 2 == 0 else k


class GaussianBlur(_Kernel):
    def forward(self, image, mask=None, kernel_size=None):
        kernel_size = kernel_size or self.get_random_kernel_size()
        return gaussian_blur(image, kernel_size), mask


class MedianFilter(_Kernel):
    def forward(self, image, mask=None, kernel_size=None):
        kernel_size = kernel_size or self.get_random_kernel_size()
        if self.passthrough:                                   # valuemetric.py:89-90
            return _ste(image, lambda t: median_filter(t, kernel_size)), mask
        return _no_adjoint(median_filter, "MedianFilter(passthrough=False)")(image, kernel_size), mask


class JPEG(_Aug):
    def __init__(self, min_quality=None, max_quality=None, passthrough=True):
        super().__init__()
        self.min_quality, self.max_quality, self.passthrough = min_quality, max_quality, passthrough

    def get_random_quality(self):
        if self.min_quality is None or self.max_quality is None:
            raise ValueError("Quality range must be specified")
        return torch.randint(self.min_quality, self.max_quality + 1, size=(1,)).item()

    def forward(self, image, mask=None, quality=None):
        quality = quality or self.get_random_quality()
        squeeze = image.dim() == 3
        img = image[None] if squeeze else image
        if self.passthrough:                                   # valuemetric.py:33-35, 41: clamp, then the straight-through estimator
            out = _ste(img, lambda t: jpeg_compress(t, quality), clamp01=True)
        else:
            out = _no_adjoint(jpeg_compress, "JPEG(passthrough=False)")(img, quality)
        return (out[0] if squeeze else out), mask


class GaussianNoise(_Aug):
    """valuemetric.py:176-194: image + randn_like(image) * std.  The normal draw is torch's own generator on the image's device,
    exactly as in the reference; the scale-and-add is the HIP kernel."""

    def __init__(self, min_std=None, max_std=None):
        super().__init__()
        self.min_std, self.max_std = min_std, max_std

    def get_random_std(self):
        if self.min_std is None or self.max_std is None:
            raise ValueError("Standard deviation range must be specified")
        return torch.rand(1).item() * (self.max_std - self.min_std) + self.min_std

    def forward(self, image, mask=None, std=None):
        std = self.get_random_std() if std is None else std
        def add(x):
            x = _dev(x)
            noise = torch.randn_like(x)
            out = torch.empty_like(x)
            N.check(N.lib().vs_aug_add_scaled(N.ptr(x), N.ptr(noise), float(std), N.ptr(out), x.numel(), N.stream()), "vs_aug_add_scaled")
            return out
        return _ste(image, add), mask                          # d(x + noise * std) / dx = 1


def gather_frames(x: torch.Tensor, indices) -> torch.Tensor:
    """frames[indices] for whole frames (any trailing shape) on the HIP gather kernel."""
    if AG.needs_grad(x):
        return AG.GatherFramesFn.apply(x, indices)
    x = _dev(x)
    idx = torch.as_tensor(indices, dtype=torch.int32).to(x.device)
    out = torch.empty((idx.numel(),) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    if idx.numel():
        N.check(N.lib().vs_aug_gather_frames(N.ptr(x), N.ptr(idx), N.ptr(out), idx.numel(), x[0].numel(), N.stream()), "vs_aug_gather_frames")
    return out


class DropFrame(_Aug):
    """augmentation/video.py:491-529: every frame is replaced by a neighbour with probability drop_frame_prob (python `random`,
    same draw order as the reference)."""

    def __init__(self, drop_frame_prob=0.125):
        super().__init__()
        self.drop_frame_prob = drop_frame_prob

    def get_random_drop_prob(self):
        return self.drop_frame_prob

    def forward(self, frames, mask=None, drop_prob=None, *args, **kwargs):
        import random
        drop_prob = drop_prob if drop_prob is not None else self.drop_frame_prob
        n = len(frames)
        idx = list(range(n))
        for i in range(n):
            if random.random() >= drop_prob:
                continue
            diff_ = -1 if random.random() < 0.5 else 1
            idx[i] = (i + diff_) % n
        return gather_frames(frames, idx), mask

    def __repr__(self):
        return f"DropFrame(prob={self.drop_frame_prob})"


class SpeedChange(_Aug):
    """augmentation/video.py:263-316: frames duplicated (speed < 1) or skipped (speed > 1) by rounded linspace indices."""

    def __init__(self, min_speed=0.5, max_speed=1.5):
        super().__init__()
        self.min_speed, self.max_speed = min_speed, max_speed

    def get_random_speed(self):
        import random
        if self.min_speed is None or self.max_speed is None:
            raise ValueError("min_speed and max_speed must be provided")
        return random.uniform(self.min_speed, self.max_speed)

    def forward(self, frames, mask=None, speed_factor=None, *args, **kwargs):
        n = frames.shape[0]
        speed_factor = speed_factor if speed_factor is not None else self.get_random_speed()
        if speed_factor == 1.0:
            return frames, mask
        if speed_factor < 1.0:
            indices = torch.linspace(0, n - 1, int(n / speed_factor)).round().long().clamp(0, n - 1)
        else:
            indices = torch.linspace(0, n - 1, int(n * speed_factor))[:n].round().long().clamp(0, n - 1)
        return gather_frames(frames, indices), (gather_frames(mask, indices) if mask is not None else None)

    def __repr__(self):
        return f"SpeedChange(min_speed={self.min_speed}, max_speed={self.max_speed})"


class TemporalReorder(_Aug):
    """augmentation/video.py:319-408: neighbouring chunks of frames swapped with probability reorder_prob (python `random`, same draw
    order as the reference); frames past the last whole chunk stay in place."""

    def __init__(self, min_chunk_size=2, max_chunk_size=5, reorder_prob=0.5):
        super().__init__()
        self.min_chunk_size, self.max_chunk_size, self.reorder_prob = min_chunk_size, max_chunk_size, reorder_prob

    def get_random_chunk_size(self):
        import random
        if self.min_chunk_size is None or self.max_chunk_size is None:
            raise ValueError("min_chunk_size and max_chunk_size must be provided")
        return random.randint(self.min_chunk_size, self.max_chunk_size)

    def forward(self, frames, mask=None, chunk_size=None, swap_probability=None, *args, **kwargs):
        import random
        n = frames.shape[0]
        chunk_size = chunk_size if chunk_size is not None else self.get_random_chunk_size()
        swap_probability = swap_probability if swap_probability is not None else self.reorder_prob
        if n < chunk_size * 2:
            return frames, mask
        nch = n // chunk_size
        order = list(range(nch))
        for i in range(0, nch - 1, 2):
            if random.random() < swap_probability and i + 1 < nch:
                order[i], order[i + 1] = order[i + 1], order[i]
        idx = [c * chunk_size + j for c in order for j in range(chunk_size)] + list(range(nch * chunk_size, n))
        return gather_frames(frames, idx), (gather_frames(mask, idx) if mask is not None else mask)

    def __repr__(self):
        return f"TemporalReorder(min_chunk_size={self.min_chunk_size}, max_chunk_size={self.max_chunk_size}, reorder_prob={self.reorder_prob})"


def window_average(frames: torch.Tensor, window_size: int, alpha: float) -> torch.Tensor:
    if AG.needs_grad(frames):
        return AG.WindowAverageFn.apply(frames, window_size, alpha)
    x = _dev(frames)
    out = torch.empty_like(x)
    N.check(N.lib().vs_aug_window_average(N.ptr(x), N.ptr(out), x.shape[0], x[0].numel(), int(window_size) // 2, float(alpha), N.stream()),
            "vs_aug_window_average")
    return out


class WindowAveraging(_Aug):
    """augmentation/video.py:411-486: every frame blended with the mean of the frames inside a sliding window."""

    def __init__(self, min_window_size=2, max_window_size=5, min_alpha=0.3, max_alpha=0.7):
        super().__init__()
        self.min_window_size, self.max_window_size, self.min_alpha, self.max_alpha = min_window_size, max_window_size, min_alpha, max_alpha

    def get_random_window_size(self):
        import random
        if self.min_window_size is None or self.max_window_size is None:
            raise ValueError("min_window_size and max_window_size must be provided")
        return random.randint(self.min_window_size, self.max_window_size)

    def get_random_alpha(self):
        import random
        if self.min_alpha is None or self.max_alpha is None:
            raise ValueError("min_alpha and max_alpha must be provided")
        return random.uniform(self.min_alpha, self.max_alpha)

    def forward(self, frames, mask=None, window_size=None, alpha=None, *args, **kwargs):
        n = frames.shape[0]
        if n <= self.min_window_size:
            return frames, mask
        window_size = window_size if window_size is not None else self.get_random_window_size()
        window_size = min(window_size, n)
        alpha = alpha if alpha is not None else self.get_random_alpha()
        return window_average(frames, window_size, alpha), mask

    def __repr__(self):
        return f"WindowAveraging(min_window={self.min_window_size}, max_window={self.max_window_size})"


class _NotBuilt(_Aug):
    why = ""

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError(f"{self.__class__.__name__}: {self.why}")


def _rotate_matrix(angle: float) -> List[float]:
    """torchvision F.rotate -> _get_inverse_affine_matrix([0,0], -angle, [0,0], 1.0, [0,0]) (python floats)."""
    rot = math.radians(-angle)
    a, b, c, d_ = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    return [d_, -b, 0.0, -c, a, 0.0]


def _affine_out_size(m: List[float], w: int, h: int) -> Tuple[int, int]:
    """torchvision _compute_affine_output_size (expand=True), float32 like the tensor code."""
    import numpy as np
    pts = np.array([[-0.5 * w, -0.5 * h, 1.0], [-0.5 * w, 0.5 * h, 1.0], [0.5 * w, 0.5 * h, 1.0], [0.5 * w, -0.5 * h, 1.0]], dtype=np.float32)
    theta = np.array(m, dtype=np.float32).reshape(2, 3)
    new = pts @ theta.T
    mn, mx = new.min(0) + np.float32([w * 0.5, h * 0.5]), new.max(0) + np.float32([w * 0.5, h * 0.5])
    tol = np.float32(1e-4)
    cmax = np.ceil(np.trunc(mx / tol) * tol)
    cmin = np.floor(np.trunc(mn / tol) * tol)
    size = cmax - cmin
    return int(size[0]), int(size[1])


def rotate(x: torch.Tensor, angle: float, expand: bool = False) -> torch.Tensor:
    """torchvision F.rotate(img, angle, interpolation=NEAREST, expand=expand, fill=None) on vs_aug_warp."""
    import numpy as np
    x = _dev(x)
    planes, H, W = _planes(x)
    m = _rotate_matrix(angle)
    ow, oh = _affine_out_size(m, W, H) if expand else (W, H)
    th = np.array(m, dtype=np.float32).reshape(2, 3)
    resc = (th.T / np.array([0.5 * W, 0.5 * H], dtype=np.float32)).astype(np.float32)      # [3][2], as _gen_affine_grid
    cl = [float(resc[k][0]) for k in range(3)] + [float(resc[k][1]) for k in range(3)]
    if AG.needs_grad(x):
        return AG.WarpFn.apply(x, 0, cl, 0, (oh, ow))
    coeffs = (C.c_float * 6)(*cl)
    out = torch.empty(x.shape[0], x.shape[1], oh, ow, device=x.device, dtype=torch.float32)
    N.check(N.lib().vs_aug_warp(N.ptr(x), N.ptr(out), planes, H, W, oh, ow, 0, coeffs, 0, N.stream()), "vs_aug_warp")
    return out


def perspective_coeffs(startpoints, endpoints) -> List[float]:
    """torchvision _get_perspective_coeffs: least squares in float64, result cast to float32."""
    a = torch.zeros(8, 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]])
        a[2 * i + 1, :] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]])
    b = torch.tensor(startpoints, dtype=torch.float64).view(8)
    res = torch.linalg.lstsq(a, b, driver="gels").solution.to(torch.float32)
    return res.tolist()


def perspective(x: torch.Tensor, startpoints, endpoints) -> torch.Tensor:
    """torchvision F.perspective(img, startpoints, endpoints, interpolation=BILINEAR, fill=None) on vs_aug_warp."""
    x = _dev(x)
    planes, H, W = _planes(x)
    pc = perspective_coeffs(startpoints, endpoints)
    if AG.needs_grad(x):
        return AG.WarpFn.apply(x, 1, pc, 1, (H, W))
    coeffs = (C.c_float * 8)(*pc)
    out = torch.empty_like(x)
    N.check(N.lib().vs_aug_warp(N.ptr(x), N.ptr(out), planes, H, W, H, W, 1, coeffs, 1, N.stream()), "vs_aug_warp")
    return out


rotate_values, perspective_values = rotate, perspective


class Rotate(_Aug):
    """geometric.py:28-59: multiples of 90 degrees with expand=True, the remainder with expand=False (both nearest)."""

    def __init__(self, min_angle=None, max_angle=None, do90=False):
        super().__init__()
        self.min_angle, self.max_angle = min_angle, max_angle
        self.base_angles = torch.tensor([-90, 0, 0, 90]) if do90 else torch.tensor([0])

    def get_random_angle(self):
        if self.min_angle is None or self.max_angle is None:
            raise ValueError("min_angle and max_angle must be provided")
        base_angle = self.base_angles[torch.randint(0, len(self.base_angles), size=(1,))].item()
        return base_angle + torch.randint(self.min_angle, self.max_angle + 1, size=(1,)).item()

    def forward(self, image, mask=None, angle=None):
        angle = angle or self.get_random_angle()
        base_angle = angle // 90 * 90
        angle = angle - base_angle
        image = rotate(rotate(image, base_angle, expand=True), angle)
        mask = rotate(rotate(mask, base_angle, expand=True), angle) if mask is not None else mask
        return image, mask


class Perspective(_Aug):
    """geometric.py:127-183."""

    def __init__(self, min_distortion_scale=None, max_distortion_scale=None):
        super().__init__()
        self.min_distortion_scale, self.max_distortion_scale = min_distortion_scale, max_distortion_scale

    def get_random_distortion_scale(self):
        if self.min_distortion_scale is None or self.max_distortion_scale is None:
            raise ValueError("min_distortion_scale and max_distortion_scale must be provided")
        return self.min_distortion_scale + torch.rand(1).item() * (self.max_distortion_scale - self.min_distortion_scale)

    def forward(self, image, mask=None, distortion_scale=None):
        distortion_scale = distortion_scale or self.get_random_distortion_scale()
        width, height = image.shape[-1], image.shape[-2]
        startpoints, endpoints = self.get_perspective_params(width, height, distortion_scale)
        image = perspective(image, startpoints, endpoints)
        mask = perspective(mask, startpoints, endpoints) if mask is not None else mask
        return image, mask

    @staticmethod
    def get_perspective_params(width, height, distortion_scale):
        half_height, half_width = height // 2, width // 2
        ri = lambda lo, hi: int(torch.randint(lo, hi, size=(1,)).item())   # noqa: E731  (same draw order as the reference)
        topleft = [ri(0, int(distortion_scale * half_width) + 1), ri(0, int(distortion_scale * half_height) + 1)]
        topright = [ri(width - int(distortion_scale * half_width) - 1, width), ri(0, int(distortion_scale * half_height) + 1)]
        botright = [ri(width - int(distortion_scale * half_width) - 1, width), ri(height - int(distortion_scale * half_height) - 1, height)]
        botleft = [ri(0, int(distortion_scale * half_width) + 1), ri(height - int(distortion_scale * half_height) - 1, height)]
        startpoints = [[0, 0], [width - 1, 0], [width - 1, height - 1], [0, height - 1]]
        return startpoints, [topleft, topright, botright, botleft]


def h264_proxy(frames: torch.Tensor, crf: int, rgb_mode: bool = False) -> torch.Tensor:
    """[F,3,H,W] -> H.264-style transform-coding proxy at QP = clamp(crf, 0, 51) (csrc/h264_proxy.hip, oracle/h264_proxy.py)."""
    x = _dev(frames)
    F_, _, H, W = x.shape
    out = torch.empty_like(x)
    L = N.lib()
    ws = torch.empty(int(L.vs_h264_proxy_workspace_bytes(F_, H, W)), dtype=torch.uint8, device=x.device)
    N.check(L.vs_h264_proxy_roundtrip(N.ptr(x), N.ptr(out), F_, H, W, int(min(max(int(crf), 0), 51)), int(rgb_mode), N.ptr(ws), N.stream()),
            "vs_h264_proxy_roundtrip")
    return out


class VideoCompression(_Aug):
    _warned = False
    """augmentation/video.py:20-119.  The reference encodes + decodes the clip with libx264 / libx265 through PyAV on the CPU and
    returns it behind a straight-through estimator.  Two back-ends here:
      * 'proxy' (default): the on-GPU transform-coding proxy of csrc/h264_proxy.hip -- H.264 4x4 core transform + quantisation at
        QP = crf, 4:2:0, no prediction / deblocking / rate control.  A documented stand-in (SURVEY.md 8(f)2), NOT libx264 parity.
      * 'pyav' (VIDEOSEAL_CODEC=pyav, needs the `av` package): the reference's own CPU round trip, kept at the boundary as a side
        path (frames leave the device exactly as in video.py:106-116).
    Forward-only, like every augmentation here (the reference's STE forward value is the codec output)."""

    def __init__(self, codec="libx264", crf=28, fps=24):
        super().__init__()
        self.codec, self.crf, self.fps = codec, crf, fps
        self.pix_fmt = "yuv420p" if codec != "libx264rgb" else "rgb24"
        self.backend = os.environ.get("VIDEOSEAL_CODEC", "proxy")
        if self.backend not in ("proxy", "pyav"):
            raise ValueError(f"VIDEOSEAL_CODEC={self.backend!r}: expected 'proxy' or 'pyav'")

    def _pyav_roundtrip(self, frames: torch.Tensor, crf: int) -> torch.Tensor:
        try:
            import av
        except ImportError as e:       # loud, never a silent fall-back to the proxy
            raise N.NativeError("VIDEOSEAL_CODEC=pyav needs the PyAV package (`av`); unset it to use the on-GPU proxy") from e
        import io
        import numpy as np
        arr = (frames.clamp(0, 1).permute(0, 2, 3, 1) * 255).to(torch.uint8).cpu().numpy()
        buf = io.BytesIO()
        with av.open(buf, mode="w", format="mp4") as box:
            st = box.add_stream(self.codec, rate=self.fps)
            st.width, st.height, st.pix_fmt = arr.shape[2], arr.shape[1], self.pix_fmt
            st.options = {"crf": str(crf), "threads": "10", "x265-params": "log_level=none"}
            for fr in arr:
                for pkt in st.encode(av.VideoFrame.from_ndarray(fr, format="rgb24")):
                    box.mux(pkt)
            for pkt in st.encode():
                box.mux(pkt)
        buf.seek(0)
        with av.open(buf, mode="r") as box:
            dec = [f.to_ndarray(format="rgb24") for f in box.decode(video=0)]
        return (torch.tensor(np.stack(dec) / 255, dtype=torch.float32).permute(0, 3, 1, 2)).to(frames.device)

    def forward(self, frames, mask=None, crf=None):
        self.crf = crf or self.crf
        if frames.shape[2] % 2 or frames.shape[3] % 2:          # video.py:98-102: pad odd sizes with zeros to even
            frames = F.pad(frames, (0, frames.shape[3] % 2, 0, frames.shape[2] % 2))
            if mask is not None:
                mask = F.pad(mask, (0, mask.shape[3] % 2, 0, mask.shape[2] % 2))
        if self.backend == "pyav":
            return _ste(frames, lambda t: self._pyav_roundtrip(t, self.crf)), mask            # video.py:113
        if not VideoCompression._warned:
            VideoCompression._warned = True
            warnings.warn(f"{self.codec}: the on-GPU transform-coding PROXY stands in for the real codec (no prediction, deblocking or rate "
                          f"control: its distortion at a given crf is not libx264's).  VIDEOSEAL_CODEC=pyav selects the reference's PyAV round trip.")
        return _ste(frames, lambda t: h264_proxy(t, self.crf, rgb_mode=(self.pix_fmt == "rgb24"))), mask

    @property
    def aug_name(self) -> str:
        """name reported in `selected_aug`: the class name, exactly like the reference (train.py builds image file names and log keys from it)"""
        return self.__class__.__name__

    @property
    def backend_name(self) -> str:
        """which implementation produces the distortion: '<Class>proxy' for the on-GPU transform-coding stand-in, the class name for PyAV"""
        return self.__class__.__name__ + ("proxy" if self.backend != "pyav" else "")

    def __repr__(self):
        return f"Compressor(codec={self.codec}, crf={self.crf}, fps={self.fps}, backend={self.backend})"


class _CrfCodec(VideoCompression):
    """video.py:147-205: H264 / H264rgb / H265 draw an integer crf in [min_crf, max_crf] from torch's global generator."""
    CODEC = "libx264"

    def __init__(self, min_crf=None, max_crf=None, fps=24):
        super().__init__(codec=self.CODEC, fps=fps)
        self.min_crf, self.max_crf = min_crf, max_crf

    def get_random_crf(self):
        if self.min_crf is None or self.max_crf is None:
            raise ValueError("min_crf and max_crf must be provided")
        return torch.randint(self.min_crf, self.max_crf + 1, size=(1,)).item()

    def forward(self, frames, mask=None, crf=None):
        return super().forward(frames, mask, crf or self.get_random_crf())

    def __repr__(self):          # (the evaluation tables and logs print this: the stand-in stays visible there; `selected_aug` is the class name)
        return self.backend_name


class H264(_CrfCodec):
    CODEC = "libx264"


class H264rgb(_CrfCodec):
    CODEC = "libx264rgb"


class H265(_CrfCodec):
    """the proxy has one transform size: at equal crf it produces the same distortion as H264's (HEVC's larger transforms are not modelled)"""
    CODEC = "libx265"


class VP9(VideoCompression):
    """video.py:208-220 (libvpx-vp9 at its default quality).  The transform-coding proxy models H.264's 4x4 transform only: VP9 / AV1 exist
    for the evaluation tables and run through the PyAV side path (VIDEOSEAL_CODEC=pyav); with the proxy back-end they raise."""

    def __init__(self, fps=24):
        super().__init__(codec="libvpx-vp9", fps=fps)
        self.crf = -1

    def forward(self, frames, mask=None, *args, **kwargs):
        if self.backend != "pyav":
            raise NotImplementedError(f"{self.codec}: no on-GPU proxy for this codec; VIDEOSEAL_CODEC=pyav runs the reference's PyAV round trip")
        return super().forward(frames, mask)

    def __repr__(self):
        return "VP9"


class AV1(_CrfCodec):
    """video.py:223-241 (libsvtav1)."""
    CODEC = "libsvtav1"

    def forward(self, frames, mask=None, crf=None):
        if self.backend != "pyav":
            raise NotImplementedError(f"{self.codec}: no on-GPU proxy for this codec; VIDEOSEAL_CODEC=pyav runs the reference's PyAV round trip")
        return super().forward(frames, mask, crf)

    def __repr__(self):
        return "AV1"


class Sequential(nn.Module):
    """augmentation/sequential.py:8-30."""

    def __init__(self, *args):
        super().__init__()
        self.transforms = args

    def forward(self, image, mask, args):
        args = tuple(args) + (None,) * (len(self.transforms) - len(args))
        items = list(zip(self.transforms, args))
        k = 0
        while k < len(items):
            transform, aug_arg = items[k]
            # round 5: a run [Crop] [Resize] [Brightness | Contrast | Saturation | Hue ...] goes through the fused passes (crop_resize_color /
            # color_chain: same values, a third of the HBM traffic of the separate launches).  Every member's parameters are drawn by its own
            # plan() in order -- the random stream is what the one-by-one calls would consume.
            if FUSE and type(transform) in (Crop, Resize, Brightness, Contrast, Saturation, Hue) and not AG.needs_grad(image) and image.is_cuda \
                    and image.dim() == 4 and image.shape[1] == 3:
                shape = tuple(image.shape)
                crop = size = None
                ops = []
                while k < len(items) and type(items[k][0]) in (Crop, Resize, Brightness, Contrast, Saturation, Hue):
                    t, a = items[k]
                    if isinstance(t, Crop):
                        if crop is not None or size is not None or ops:
                            break
                        kind, crop = t.plan(shape, a)
                        shape = shape[:-2] + (crop[2], crop[3])
                    elif isinstance(t, Resize):
                        if size is not None or ops:
                            break
                        kind, size = t.plan(shape, a)
                        shape = shape[:-2] + tuple(size)
                    else:
                        ops.append(t.plan(shape, a)[1])
                    k += 1
                if size is not None:
                    image = crop_resize_color(image, crop, size, ops)
                    if mask is not None:
                        mask = resize(crop_flip(mask, *crop) if crop is not None else mask, size, True)
                else:
                    if crop is not None:
                        image = crop_flip(image, *crop)
                        if mask is not None:
                            mask = crop_flip(mask, *crop)
                    image = color_chain(image, ops)
                continue
            image, mask = transform(image, mask, aug_arg)
            k += 1
        return image, mask

    def __repr__(self):
        return f"{self.transforms}".replace(", ", "_")


name2aug = {"resize": Resize, "crop": Crop, "hflip": HorizontalFlip, "identity": Identity, "jpeg": JPEG, "gaussian_blur": GaussianBlur,
            "median_filter": MedianFilter, "brightness": Brightness, "contrast": Contrast, "saturation": Saturation, "hue": Hue,
            "rotate": Rotate, "perspective": Perspective, "h264": H264, "h264rgb": H264rgb, "h265": H265, "video_compression": VideoCompression,
            "drop_frame": DropFrame, "gaussian_noise": GaussianNoise, "grayscale": Grayscale, "speed_change": SpeedChange}
video_augs = ["video_compression", "h264", "h264rgb", "h265"]


class NoMaskEmbedder:
    """augmentation/masks.py:305-314: the whole frame is watermarked (configs/all_augs.yaml:2-3 `kind: none`)."""

    def __call__(self, imgs, iter_i=None, raw_image=None, **kwargs):
        return torch.ones_like(imgs[:, 0:1, ...])

    def sample_representative_masks(self, img):
        return torch.ones((1, 1, img.shape[-2], img.shape[-1]))


class GivenMaskEmbedder:
    """The `masks` the caller passes to forward() are the mask targets (the CocoSegmentation branch of masks.py:395-396)."""

    def __call__(self, imgs, masks=None, **kwargs):
        if masks is None:
            raise ValueError("GivenMaskEmbedder needs the masks argument of forward()")
        return masks


class _MixedMaskEmbedder:
    why = ("the 'mixed' mask embedder (masks.py:317-424) draws irregular strokes / boxes with OpenCV + numpy on the host and is out of "
           "scope (SURVEY.md section 2); use masks={'kind': 'none'} (the training config) or assign any callable "
           "`augmenter.mask_embedder = fn(imgs_w, masks=...) -> [F,1,H,W]`")

    def __call__(self, *a, **k):
        raise NotImplementedError(self.why)


def get_mask_embedder(kind=None, **kwargs):
    """augmentation/masks.py:426-438."""
    if kind is None:
        kind = "mixed"
    if kind == "none":
        return NoMaskEmbedder()
    if kind == "given":
        return GivenMaskEmbedder()
    if kind == "mixed":
        return _MixedMaskEmbedder()
    raise NotImplementedError(f"No such embedder kind = {kind}")


def mask_blend(imgs_w: torch.Tensor, imgs: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """augmenter.py:175  imgs_w * m + imgs * (1 - m), m [F,1,H,W]."""
    if AG.needs_grad(imgs_w, imgs):
        return AG.MaskBlendFn.apply(imgs_w, imgs, mask)
    a, b, m = _dev(imgs_w), _dev(imgs), _dev(mask)
    F_, Cc, H, W = a.shape
    if b.shape != a.shape or tuple(m.shape) != (F_, 1, H, W):
        raise ValueError(f"mask blend: shapes {tuple(a.shape)}, {tuple(b.shape)}, {tuple(m.shape)}")
    out = torch.empty_like(a)
    N.check(N.lib().vs_aug_mask_blend(N.ptr(a), N.ptr(b), N.ptr(m), N.ptr(out), F_, Cc, H, W, N.stream()), "vs_aug_mask_blend")
    return out


class Augmenter(nn.Module):
    """augmentation/augmenter.py:60-199."""

    def __init__(self, masks: dict, augs: dict, augs_params: dict, num_augs: int = 1, **kwargs) -> None:
        super().__init__()
        self.mask_embedder = get_mask_embedder(**(masks or {}))
        self.augs, self.aug_probs = self.parse_augmentations(augs, augs_params)
        self.augs_video, self.aug_probs_video = self.parse_augmentations(augs, augs_params, is_video=True)
        self.num_augs = num_augs

    @staticmethod
    def parse_augmentations(augs: Dict[str, float], augs_params: Dict[str, dict], is_video: bool = False):
        out, probs = [], []
        for name in augs.keys():
            if name in video_augs and not is_video:
                continue
            if name not in name2aug:
                raise ValueError(f"Augmentation {name} not found. Add it in name2aug.")
            out.append(name2aug[name](**(augs_params.get(name, {}) or {})))
            probs.append(float(augs[name]))
        total = sum(probs)
        return nn.ModuleList(out), torch.tensor([p / total for p in probs])

    def augment(self, image, mask, is_video, do_resize=True):
        augs = self.augs_video if is_video else self.augs
        probs = self.aug_probs_video if is_video else self.aug_probs
        aug = augs[torch.multinomial(probs, 1).item()]
        h, w = image.shape[-2:]
        image, mask = aug(image, mask)
        if do_resize and image.shape[-2:] != (h, w):
            image = resize(image, (h, w), True)
            mask = resize(mask, (h, w), True)
        return image, mask, getattr(aug, "aug_name", aug.__class__.__name__)

    def forward(self, imgs_w, imgs, masks, is_video=True, do_resize=True):
        """augmenter.py:154-194.  Training: mask targets from the mask embedder, imgs_w * m + imgs * (1 - m), then num_augs picks.
        Eval: the reference's branch references an unassigned variable (augmenter.py:185-194 cannot run); here it is the
        full-mask case, i.e. the picks applied to imgs_w."""
        if self.training and not isinstance(self.mask_embedder, NoMaskEmbedder):
            mask_targets = self.mask_embedder(imgs_w, masks=masks).to(imgs_w.device)
            imgs_aug = mask_blend(imgs_w, imgs, mask_targets.float().expand(imgs_w.shape[0], 1, *imgs_w.shape[-2:]).contiguous())
        else:           # m = 1: imgs_w * 1 + imgs * 0 is imgs_w itself, no pass over the frames
            mask_targets = torch.ones_like(imgs_w)[:, 0:1]
            imgs_aug = imgs_w
        names: List[str] = []
        for _ in range(self.num_augs):
            imgs_aug, mask_targets, nm = self.augment(imgs_aug, mask_targets, is_video, do_resize)
            names.append(nm)
        return imgs_aug, mask_targets, "+".join(names)


def get_dummy_augmenter():
    """augmenter.py:48-57 (full mask instead of the OpenCV 'mixed' embedder, see model.get_dummy_augmenter)."""
    return Augmenter(augs={"identity": 1}, augs_params={}, masks={"kind": "none"})


def get_validation_augs_subset(is_video: bool = False) -> list:
    """augmentation/__init__.py:12-40."""
    codec, q = (H264, 40) if is_video else (JPEG, 60)
    return [(Identity(), [0]), (HorizontalFlip(), [0]), (Crop(), [0.71]), (Brightness(), [0.5]), (codec(), [q]),
            (Sequential(codec(), Crop(), Brightness()), [(q, 0.71, 0.5)])]


def get_combined_augs(is_video: bool = False) -> list:
    """augmentation/__init__.py:43-59."""
    if is_video:
        return [(Identity(), [0]), (Sequential(H264(), Crop(), Brightness()), [(30, 0.71, 0.5)]),
                (Sequential(H264(), Crop(), Brightness()), [(40, 0.71, 0.5)])]
    return [(Identity(), [0]), (Sequential(JPEG(), Crop(), Brightness()), [(40, 0.71, 0.5)])]


def get_validation_augs(is_video: bool = False, only_identity: bool = False, only_combined: bool = False) -> list:
    """The fixed-strength evaluation tables of augmentation/__init__.py:62-130, row for row.  The codec rows run on the back-end selected by
    VIDEOSEAL_CODEC (default: the on-GPU H.264 proxy, reported as `H264proxy`; VP9 needs the PyAV side path)."""
    if only_identity:
        return [(Identity(), [0])]
    if only_combined:
        return get_combined_augs(is_video)
    if is_video:
        return [(Identity(), [0]), (HorizontalFlip(), [0]), (Rotate(), [10, 90]), (Resize(), [0.55, 0.71]), (Crop(), [0.55, 0.71]),
                (Perspective(), [0.5]), (Brightness(), [0.5, 1.5]),
                (Contrast(), [0.5, 1.5]), (Saturation(), [0.5, 1.5]), (Hue(), [0.25]), (Grayscale(), [-1]), (JPEG(), [40]), (GaussianBlur(), [9]),
                (H264(), [23, 30, 40, 50]), (H264rgb(), [23, 30, 40, 50]), (H265(), [23, 30, 40, 50]), (VP9(), [-1]),
                (Sequential(H264(), Crop(), Brightness()), [(23, 0.71, 0.5)]), (Sequential(H264(), Crop(), Brightness()), [(30, 0.71, 0.5)]),
                (Sequential(H264(), Crop(), Brightness()), [(40, 0.71, 0.5)]), (Sequential(H264(), Crop(), Brightness()), [(50, 0.71, 0.5)])]
    return [(Identity(), [0]), (HorizontalFlip(), [0]), (Rotate(), [5, 10, 30, 45, 90]),
            (Resize(), [0.32, 0.45, 0.55, 0.63, 0.71, 0.77, 0.84, 0.89, 0.95, 1.00]),
            (Crop(), [0.32, 0.45, 0.55, 0.63, 0.71, 0.77, 0.84, 0.89, 0.95, 1.00]),
            (Perspective(), [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]),
            (Brightness(), [0.1, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0]), (Contrast(), [0.1, 0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75, 2.0]),
            (Hue(), [-0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5]), (Grayscale(), [-1]), (JPEG(), [40, 50, 60, 70, 80, 90]),
            (GaussianBlur(), [3, 5, 9, 13, 17]),
            (Sequential(JPEG(), Crop(), Brightness()), [(40, 0.71, 0.5)]), (Sequential(JPEG(), Crop(), Brightness()), [(60, 0.71, 0.5)]),
            (Sequential(JPEG(), Crop(), Brightness()), [(80, 0.71, 0.5)])]
