"""Independent pins for the torchvision-dependent augmentation oracle (oracle/augment.py).

torchvision is neither installed here nor vendored by the reference, so the oracle restates `torchvision.transforms.functional`
(SURVEY.md appendix C).  These tests check that restatement against implementations that do NOT come from this repository:
  * Pillow's ImageEnhance / ImageOps -- the functions torchvision's own PIL backend calls for adjust_brightness / contrast /
    saturation (`_functional_pil.py`), i.e. the semantics the tensor backend is specified to reproduce up to 8-bit rounding;
  * the standard library's colorsys for the RGB <-> HSV arithmetic of adjust_hue;
  * scipy.ndimage for the Gaussian blur (same sampled-Gaussian kernel, mirror padding), the nearest-neighbour rotation and the
    bilinear perspective warp (independent interpolation engines, numpy.linalg for the homography).
They run on the CPU (`-m "not gpu"`).  The GPU kernels are compared with the same oracle in tests/test_gpu_aug.py."""
import colorsys

import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance
from scipy import ndimage

from oracle import augment as A
from oracle.inputs import synthetic_frames


def _u8_frames(n=2, h=48, w=64, seed=7):
    x = synthetic_frames(n, h, w, seed=seed)
    return (x * 255).round().clamp(0, 255).to(torch.uint8)


def _pil(u8):          # [3,H,W] uint8 -> PIL RGB
    return Image.fromarray(u8.permute(1, 2, 0).numpy(), "RGB")


@pytest.mark.parametrize("name,enh,factors", [
    ("brightness", ImageEnhance.Brightness, [0.5, 0.8, 1.25, 1.5, 2.0]),
    ("contrast", ImageEnhance.Contrast, [0.5, 0.8, 1.25, 1.5, 2.0]),
    ("saturation", ImageEnhance.Color, [0.0, 0.5, 1.5, 2.0]),
])
def test_colour_blends_match_pillow_enhancers(name, enh, factors):
    """oracle (float blend + clamp) vs the Pillow enhancer torchvision's PIL backend uses: equal up to the 8-bit rounding of the
    Pillow path (its degenerate image -- black / rounded mean grey / L conversion -- is itself an 8-bit image)."""
    u8 = _u8_frames()
    op = getattr(A, name)
    for f in factors:
        got = op(u8.float() / 255.0, f) * 255.0
        for i in range(u8.shape[0]):
            ref = torch.from_numpy(np.asarray(enh(_pil(u8[i])).enhance(f)).copy()).permute(2, 0, 1).float()
            diff = (got[i] - ref).abs()
            # one level of rounding in the degenerate image, scaled by |1 - f|, plus one level in the blend
            assert diff.max() <= 1.0 + abs(1.0 - f) * 1.0 + 1e-3, (name, f, diff.max())
            assert diff.mean() < 0.75


def test_hue_matches_colorsys():
    """adjust_hue = RGB -> HSV, h += f (mod 1), HSV -> RGB: against the standard library's colorsys on every pixel (float arithmetic)."""
    x = synthetic_frames(1, 24, 32, seed=3)
    for f in (-0.5, -0.1, 0.1, 0.25, 0.5):
        got = A.hue(x, f)[0]
        ref = torch.empty_like(got)
        xn = x[0].numpy()
        for yy in range(x.shape[-2]):
            for xx in range(x.shape[-1]):
                h, s, v = colorsys.rgb_to_hsv(float(xn[0, yy, xx]), float(xn[1, yy, xx]), float(xn[2, yy, xx]))
                r, g, b = colorsys.hsv_to_rgb((h + f) % 1.0, s, v)
                ref[0, yy, xx], ref[1, yy, xx], ref[2, yy, xx] = r, g, b
        assert (got - ref).abs().max() < 2e-5, f


@pytest.mark.parametrize("k", [3, 5, 9, 13, 17])
def test_gaussian_blur_matches_scipy(k):
    """torchvision.gaussian_blur(kernel_size=k, sigma=None): sampled Gaussian of sigma = 0.3 ((k-1)/2 - 1) + 0.8 on k taps, reflect
    padding -- scipy.ndimage.gaussian_filter with the same sigma, truncate = (k // 2) / sigma and mode='mirror' is the same filter."""
    x = synthetic_frames(2, 40, 56, seed=11)
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    got = A.gaussian_blur(x, k).numpy()
    ref = ndimage.gaussian_filter(x.numpy().astype(np.float64), sigma=(0, 0, sigma, sigma), truncate=(k // 2) / sigma + 1e-9, mode="mirror")
    assert np.abs(got - ref).max() < 2e-6


def test_rotate_right_angles_are_exact_and_general_angles_match_scipy():
    """F.rotate is counter-clockwise, about the image centre, nearest neighbour, zero fill.  Right angles on a square image are exact
    permutations (numpy.rot90); other angles agree with scipy.ndimage.rotate(order=0) except on pixels whose source coordinate lies
    within rounding distance of a pixel boundary."""
    x = synthetic_frames(1, 48, 48, seed=5)
    xn = x.numpy()
    assert np.array_equal(A.rotate(x, 90).numpy(), np.rot90(xn, 1, axes=(-2, -1)))
    assert np.array_equal(A.rotate(x, -90).numpy(), np.rot90(xn, -1, axes=(-2, -1)))
    assert np.array_equal(A.rotate(x, 180).numpy(), np.rot90(xn, 2, axes=(-2, -1)))
    for angle in (10, 30, -17, 45):
        got = A.rotate(x, angle).numpy()
        ref = ndimage.rotate(xn, angle, axes=(-1, -2), reshape=False, order=0, mode="constant", cval=0.0)
        frac = float((np.abs(got - ref) > 1e-6).mean())
        assert frac < 0.06, (angle, frac)          # boundary pixels of the nearest-neighbour rounding only
        # and never a different geometry: the bilinear versions of both agree closely in the interior
        inner = (slice(None), slice(None), slice(8, -8), slice(8, -8))
        sm = ndimage.uniform_filter(got, size=(1, 1, 5, 5))[inner] - ndimage.uniform_filter(ref, size=(1, 1, 5, 5))[inner]
        assert np.abs(sm).mean() < 0.01


@pytest.mark.parametrize("scale", [0.1, 0.3, 0.5])
def test_perspective_matches_an_independent_homography_warp(scale):
    """F.perspective(startpoints -> endpoints, bilinear, zero fill): output pixel (x, y) samples the input at the homography that maps
    the END points onto the START points.  Independent check: homography by numpy.linalg.solve on the 8 x 8 system, sampling by
    scipy.ndimage.map_coordinates(order=1)."""
    h, w = 40, 56
    x = synthetic_frames(1, h, w, seed=9)
    g = torch.Generator().manual_seed(3)
    half_h, half_w = h // 2, w // 2
    d = lambda hi: int(torch.randint(0, int(scale * hi) + 1, (1,), generator=g))      # noqa: E731
    start = [[0, 0], [w - 1, 0], [w - 1, h - 1], [0, h - 1]]
    end = [[d(half_w), d(half_h)], [w - 1 - d(half_w), d(half_h)], [w - 1 - d(half_w), h - 1 - d(half_h)], [d(half_w), h - 1 - d(half_h)]]
    got = A.perspective(x, start, end)[0].numpy()
    # homography H with H(end_i) = start_i
    Am, bv = [], []
    for (ex, ey), (sx, sy) in zip(end, start):
        Am.append([ex, ey, 1, 0, 0, 0, -sx * ex, -sx * ey]); bv.append(sx)
        Am.append([0, 0, 0, ex, ey, 1, -sy * ex, -sy * ey]); bv.append(sy)
    c = np.linalg.solve(np.array(Am, dtype=np.float64), np.array(bv, dtype=np.float64))
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    # torchvision evaluates the homography at pixel centres (x + 0.5, y + 0.5) and grid_sample(align_corners=False) reads at source - 0.5
    px, py = xs + 0.5, ys + 0.5
    den = c[6] * px + c[7] * py + 1.0
    sx = (c[0] * px + c[1] * py + c[2]) / den - 0.5
    sy = (c[3] * px + c[4] * py + c[5]) / den - 0.5
    ref = np.stack([ndimage.map_coordinates(x[0, ch].numpy().astype(np.float64), [sy, sx], order=1, mode="grid-constant", cval=0.0) for ch in range(3)])
    inner = (slice(None), slice(2, -2), slice(2, -2))       # (zero-fill border handling differs by construction at the outermost pixel)
    assert np.abs(got[inner] - ref[inner]).max() < 1e-4
