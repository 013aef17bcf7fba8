"""Pins of the augmentation oracle (CPU only): the integer libjpeg restatement is BIT-EXACT with Pillow."""
import numpy as np
import pytest
import torch

from oracle import augment as A
from oracle.inputs import synthetic_frames
from oracle.jpeg_ref import jpeg_roundtrip, pil_roundtrip


@pytest.mark.parametrize("hw", [(64, 64), (100, 75), (37, 53), (50, 50), (120, 200), (17, 19), (8, 8), (1, 1), (2, 3), (256, 256)])
@pytest.mark.parametrize("quality", [40, 60, 75, 90])
def test_integer_jpeg_restatement_is_bit_exact_with_pillow(hw, quality):
    H, W = hw
    for kind in ("smooth", "uniform"):
        x = (synthetic_frames(1, H, W, seed=H * 7 + W + quality, kind=kind)[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        assert np.array_equal(jpeg_roundtrip(x, quality), pil_roundtrip(x, quality)), (hw, quality, kind)


def test_colour_ops_identities():
    x = synthetic_frames(2, 24, 20, seed=1)
    assert torch.allclose(A.brightness(x, 1.0), x)
    assert torch.allclose(A.saturation(x, 1.0), x) and torch.allclose(A.contrast(x, 1.0), x)
    assert torch.allclose(A.hue(x, 0.0), x, atol=1e-6) and torch.allclose(A.hue(x, 1.0), x, atol=1e-5)
    g = A.grayscale(x)
    assert torch.equal(g[:, 0], g[:, 1]) and torch.equal(g[:, 1], g[:, 2])
    assert torch.allclose(A.saturation(x, 0.0), A.gray_tv(x).expand_as(x).clamp(0, 1))
    assert torch.equal(A.hflip(A.hflip(x)), x)


def test_median_is_median_of_row_medians():
    x = torch.rand(1, 1, 5, 5)
    m = A.median_filter(x, 3)
    win = x[0, 0, 1:4, 1:4]
    assert m[0, 0, 2, 2] == win.median(dim=-1).values.median()


def test_gaussian_blur_preserves_constants():
    x = torch.full((1, 3, 20, 20), 0.37)
    assert torch.allclose(A.gaussian_blur(x, 9), x, atol=1e-6)


def test_rotate_perspective_restatement_sanity():
    """oracle/augment.py rotate / perspective (torchvision restated; unpinned -- torchvision is not installed): closed-form cases."""
    import torch
    from oracle import augment as A
    x = torch.rand(2, 3, 12, 18)
    assert torch.equal(A.rotate(x, 90, expand=True), torch.rot90(x, 1, dims=(-2, -1)))
    assert torch.equal(A.rotate(x, -90, expand=True), torch.rot90(x, -1, dims=(-2, -1)))
    assert torch.equal(A.rotate(x, 0, expand=True), x) and torch.equal(A.rotate(x, 0), x)
    r = A.rotate(x, 30)
    assert r.shape == x.shape and (r[..., 0, 0] == 0).all()            # corners rotate out of the frame: zero fill
    sp = [[0, 0], [17, 0], [17, 11], [0, 11]]
    assert (A.perspective(x, sp, sp) - x).abs().max() < 1e-5            # identity homography
    ep = [[2, 1], [15, 2], [16, 10], [1, 9]]
    c = A.perspective_coeffs(sp, ep)
    for (sx, sy), (ex, ey) in zip(sp, ep):                              # the homography maps end points back onto start points
        den = c[6] * ex + c[7] * ey + 1
        assert abs((c[0] * ex + c[1] * ey + c[2]) / den - sx) < 1e-3 and abs((c[3] * ex + c[4] * ey + c[5]) / den - sy) < 1e-3


def test_h264_proxy_definition_is_well_behaved():
    """oracle/h264_proxy.py (the DEFINITION of the codec stand-in; libx264 itself cannot be pinned offline): a flat frame survives any
    QP exactly up to the colour conversion, distortion grows with crf, padding is cropped, rgb mode codes planes directly."""
    import numpy as np
    from oracle import h264_proxy as HP
    from oracle.inputs import synthetic_frames
    x = synthetic_frames(2, 37, 50, seed=5).numpy()
    mse = [float(((HP.roundtrip(x, q) - x) ** 2).mean()) for q in (0, 12, 23, 34, 45, 51)]
    assert mse == sorted(mse) and mse[0] < 4e-4 and mse[-1] > 10 * mse[0]
    assert HP.roundtrip(x, 28).shape == x.shape
    flat = np.full((1, 3, 16, 16), 128 / 255, dtype=np.float32)
    assert np.abs(HP.roundtrip(flat, 51, True) - flat).max() < 1e-6          # a flat 128 block has a zero residual
    assert np.abs(HP.roundtrip(flat, 51) - flat).max() <= 2 / 255            # + limited-range YCbCr rounding
    # the 4x4 transform pair is exact at QP 0 up to the quantiser's rounding: |error| <= 1 grey level in rgb mode
    assert np.abs(HP.roundtrip(x, 0, True) - np.floor(np.clip(x, 0, 1) * 255) / 255).max() <= 1 / 255 + 1e-7
