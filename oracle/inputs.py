"""Seeded synthetic frames / messages shared by tests, goldens and bench (TEST INFRASTRUCTURE).

SURVEY.md section 8(d): uniform noise alone makes JND / JPEG degenerate, so the
default content is smooth (low-pass noise + ramps + a few hard edges).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def synthetic_frames(n: int, h: int, w: int, seed: int = 0, kind: str = "smooth") -> torch.Tensor:
    """[n,3,h,w] float32 in [0,1]; consecutive frames drift slowly (video-like)."""
    g = torch.Generator().manual_seed(1234567 + seed)
    if kind == "uniform":
        return torch.rand(n, 3, h, w, generator=g)
    lo = torch.rand(1, 3, max(h // 16, 2) + n, max(w // 16, 2) + n, generator=g)
    mid = torch.rand(1, 3, max(h // 4, 2) + n, max(w // 4, 2) + n, generator=g)
    frames = []
    yy = torch.linspace(0, 1, h)[:, None].expand(h, w)
    xx = torch.linspace(0, 1, w)[None, :].expand(h, w)
    for i in range(n):
        a = F.interpolate(lo[:, :, i:i + max(h // 16, 2), i:i + max(w // 16, 2)], size=(h, w), mode="bicubic", align_corners=False)
        b = F.interpolate(mid[:, :, i:i + max(h // 4, 2), i:i + max(w // 4, 2)], size=(h, w), mode="bilinear", align_corners=False)
        img = 0.55 * a + 0.25 * b + 0.2 * torch.stack([yy, xx, 1 - yy])[None]
        # hard edges: a rectangle and a disc that move with i
        cy, cx = 0.3 + 0.02 * i, 0.6 - 0.015 * i
        disc = (((yy - cy) ** 2 + (xx - cx) ** 2) < 0.02).float()
        rect = ((yy > 0.6) & (yy < 0.8) & (xx > 0.1 + 0.01 * i) & (xx < 0.35 + 0.01 * i)).float()
        img = img * (1 - 0.5 * disc) + 0.35 * rect * torch.tensor([1.0, -0.5, 0.25])[None, :, None, None]
        frames.append(img)
    out = torch.cat(frames, 0) + 0.02 * (torch.rand(n, 3, h, w, generator=g) - 0.5)
    return out.clamp(0, 1).contiguous()


def synthetic_msgs(bsz: int, nbits: int, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(7654321 + seed)
    return torch.randint(0, 2, (bsz, nbits), generator=g)
