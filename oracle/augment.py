"""CPU restatement of the augmentations (TEST INFRASTRUCTURE, see oracle/__init__.py).

Geometric / filter ops follow videoseal/augmentation/{geometric,valuemetric}.py and utils/image.py; the colour ops call
torchvision.transforms.functional in the reference, which is NOT installed and not vendored: they are restated from the
published torchvision `_functional_tensor.py` semantics -- parity with torchvision itself is "unpinned"; the
restatement is checked against independent implementations instead (tests/test_oracle_aug_pins.py): Pillow's ImageEnhance (what
torchvision's own PIL backend calls) for brightness / contrast / saturation, the standard library's colorsys for hue, scipy.ndimage for
the Gaussian blur, the rotation and (with a numpy.linalg homography) the perspective warp.
JPEG is the Pillow round trip itself (pinned: Pillow is the reference's codec).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .jpeg_ref import pil_roundtrip


def _blend(a, b, r):
    return (r * a + (1.0 - r) * b).clamp(0, 1.0)


def gray_tv(x):
    r, g, b = x.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def brightness(x, f):
    return _blend(x, torch.zeros_like(x), f)


def contrast(x, f):
    mean = torch.mean(gray_tv(x), dim=(-3, -2, -1), keepdim=True)
    return _blend(x, mean, f)


def saturation(x, f):
    return _blend(x, gray_tv(x), f)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / cr_divisor, (maxc - g) / cr_divisor, (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * f), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def hue(x, f):
    hsv = _rgb2hsv(x)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + f) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


def grayscale(x):
    """valuemetric.py:196-208."""
    g = 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
    return g.expand_as(x).contiguous()


def hflip(x):
    return x.flip(-1)


def crop(x, i, j, h, w):
    return x[..., i:i + h, j:j + w].contiguous()


def resize(x, size, antialias=True):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False, antialias=antialias)


def gaussian_blur(x, k):
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    half = (k - 1) * 0.5
    t = torch.linspace(-half, half, steps=k)
    pdf = torch.exp(-0.5 * (t / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    C = x.shape[-3]
    pad = [k // 2] * 4
    xp = F.pad(x, pad, mode="reflect")
    return F.conv2d(xp, k2.expand(C, 1, k, k), groups=C)


def median_filter(x, k):
    """utils/image.py:60-84: median of row medians, zero padded."""
    p = k // 2
    xp = F.pad(x, (p, p, p, p))
    blocks = xp.unfold(2, k, 1).unfold(3, k, 1)
    return blocks.median(dim=-1).values.median(dim=-1).values


def jpeg(x, quality):
    """valuemetric.py:39-46 + utils/image.py:24-34: clamp, ToPILImage (mul(255).byte(): truncation), Pillow, ToTensor."""
    x = torch.clamp(x, 0, 1)
    out = torch.empty_like(x)
    for i in range(x.shape[0]):
        u8 = x[i].mul(255).byte().permute(1, 2, 0).numpy()
        out[i] = torch.from_numpy(pil_roundtrip(np.ascontiguousarray(u8), quality).copy()).permute(2, 0, 1).float() / 255
    return out


# ---- Rotate / Perspective (geometric.py:28-59, 127-183).  torchvision is neither installed nor vendored by the reference
# ("parity unpinned"): this restates torchvision/_functional_tensor.py (_gen_affine_grid, _perspective_grid,
# _compute_affine_output_size, _get_inverse_affine_matrix, _get_perspective_coeffs) with torch ops and uses the real ATen
# F.grid_sample for the sampling itself.
def _inverse_rotate_matrix(angle):
    import math
    rot = math.radians(-angle)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    return [d, -b, 0.0, -c, a, 0.0]


def _affine_output_size(matrix, w, h):
    pts = torch.tensor([[-0.5 * w, -0.5 * h, 1.0], [-0.5 * w, 0.5 * h, 1.0], [0.5 * w, 0.5 * h, 1.0], [0.5 * w, -0.5 * h, 1.0]])
    theta = torch.tensor(matrix, dtype=torch.float).view(2, 3)
    new_pts = torch.matmul(pts, theta.T)
    min_vals, _ = new_pts.min(dim=0)
    max_vals, _ = new_pts.max(dim=0)
    min_vals += torch.tensor((w * 0.5, h * 0.5))
    max_vals += torch.tensor((w * 0.5, h * 0.5))
    tol = 1e-4
    cmax = torch.ceil((max_vals / tol).trunc_() * tol)
    cmin = torch.floor((min_vals / tol).trunc_() * tol)
    size = cmax - cmin
    return int(size[0]), int(size[1])


def rotate(x, angle, expand=False):
    """F.rotate(img, angle, interpolation=NEAREST, expand=expand, center=None, fill=None)."""
    h, w = x.shape[-2:]
    matrix = _inverse_rotate_matrix(angle)
    ow, oh = _affine_output_size(matrix, w, h) if expand else (w, h)
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, oh, ow, 3)
    base[..., 0].copy_(torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow))
    base[..., 1].copy_(torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h])
    grid = base.view(1, oh * ow, 3).bmm(rescaled).view(1, oh, ow, 2)
    return F.grid_sample(x, grid.expand(x.shape[0], oh, ow, 2), mode="nearest", padding_mode="zeros", align_corners=False)


def perspective_coeffs(startpoints, endpoints):
    a = torch.zeros(2 * len(startpoints), 8, dtype=torch.float64)
    for i, (p1, p2) in enumerate(zip(endpoints, startpoints)):
        a[2 * i, :] = torch.tensor([p1[0], p1[1], 1, 0, 0, 0, -p2[0] * p1[0], -p2[0] * p1[1]])
        a[2 * i + 1, :] = torch.tensor([0, 0, 0, p1[0], p1[1], 1, -p2[1] * p1[0], -p2[1] * p1[1]])
    b = torch.tensor(startpoints, dtype=torch.float64).view(8)
    return torch.linalg.lstsq(a, b, driver="gels").solution.to(torch.float32).tolist()


def perspective(x, startpoints, endpoints):
    """F.perspective(img, startpoints, endpoints, interpolation=BILINEAR, fill=None)."""
    oh, ow = x.shape[-2:]
    c = perspective_coeffs(startpoints, endpoints)
    theta1 = torch.tensor([[[c[0], c[1], c[2]], [c[3], c[4], c[5]]]])
    theta2 = torch.tensor([[[c[6], c[7], 1.0], [c[6], c[7], 1.0]]])
    d = 0.5
    base = torch.empty(1, oh, ow, 3)
    base[..., 0].copy_(torch.linspace(d, ow * 1.0 + d - 1.0, steps=ow))
    base[..., 1].copy_(torch.linspace(d, oh * 1.0 + d - 1.0, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled1 = theta1.transpose(1, 2) / torch.tensor([0.5 * ow, 0.5 * oh])
    g1 = base.view(1, oh * ow, 3).bmm(rescaled1)
    g2 = base.view(1, oh * ow, 3).bmm(theta2.transpose(1, 2))
    grid = (g1 / g2 - 1.0).view(1, oh, ow, 2)
    return F.grid_sample(x, grid.expand(x.shape[0], oh, ow, 2), mode="bilinear", padding_mode="zeros", align_corners=False)


# ---- Augmenter (augmentation/augmenter.py:60-194) with the ops the golden forward cases use.  RNG draws in the reference's order:
# torch.multinomial(probs, 1) for the pick, then the op's own draws (Crop: randint h, randint w, then RandomCrop.get_params'
# randint i, randint j unless the size is unchanged).
class Augmenter:
    OPS = {"identity": "Identity", "crop": "Crop", "hflip": "HorizontalFlip", "resize": "Resize"}

    def __init__(self, augs: dict, augs_params: dict, num_augs: int = 1):
        self.names = list(augs.keys())
        tot = sum(float(v) for v in augs.values())
        self.probs = torch.tensor([float(augs[k]) / tot for k in self.names])
        self.params = augs_params
        self.num_augs = num_augs

    def _size(self, name, h, w):
        p = self.params[name]
        return (torch.randint(int(p["min_size"] * h), int(p["max_size"] * h) + 1, size=(1,)).item(),
                torch.randint(int(p["min_size"] * w), int(p["max_size"] * w) + 1, size=(1,)).item())

    def _apply(self, name, image, mask):
        h, w = image.shape[-2:]
        if name == "identity":
            return image, mask
        if name == "hflip":
            return hflip(image), (hflip(mask) if mask is not None else mask)
        if name == "resize":
            out = self._size(name, h, w)
            return resize(image, out), (resize(mask, out) if mask is not None else mask)
        if name == "crop":
            th, tw = self._size(name, h, w)
            if (th, tw) == (h, w):
                i = j = 0
            else:
                i = torch.randint(0, h - th + 1, size=(1,)).item()
                j = torch.randint(0, w - tw + 1, size=(1,)).item()
            return crop(image, i, j, th, tw), (crop(mask, i, j, th, tw) if mask is not None else mask)
        raise ValueError(name)

    def __call__(self, imgs_w, imgs, masks, is_video=True, do_resize=True):
        """training branch with the full mask (NoMaskEmbedder): mask_targets = 1, imgs_aug = imgs_w * 1 + imgs * 0."""
        mask_targets = torch.ones_like(imgs_w[:, 0:1])
        imgs_aug = imgs_w * mask_targets + imgs * (1 - mask_targets)
        names = []
        for _ in range(self.num_augs):
            name = self.names[torch.multinomial(self.probs, 1).item()]
            h, w = imgs_aug.shape[-2:]
            imgs_aug, mask_targets = self._apply(name, imgs_aug, mask_targets)
            if do_resize and tuple(imgs_aug.shape[-2:]) != (h, w):
                imgs_aug = resize(imgs_aug, (h, w))
                mask_targets = resize(mask_targets, (h, w))
            names.append(self.OPS[name])
        return imgs_aug, mask_targets, "+".join(names)
