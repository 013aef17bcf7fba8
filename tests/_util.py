"""Shared helpers for the parity tests (golden loading, sub-sample comparison)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def check_sub(g, prefix, t, atol, what=""):
    """compare tensor ``t`` with the stored strided sub-sample + float64 checksums."""
    stats = g[prefix + ".stats"]
    stride = int(stats[3])
    t = t.detach().cpu().float()
    assert t.numel() == int(stats[2]), f"{what}{prefix}: numel {t.numel()} != {int(stats[2])}"
    sub = t.flatten()[::stride].numpy()
    err = np.abs(sub - g[prefix + ".sub"]).max()
    assert err <= atol, f"{what}{prefix}: max abs err {err:.3e} > {atol}"
    mean_ref = stats[0] / stats[2]
    mean = float(t.double().mean())
    assert abs(mean - mean_ref) <= atol, f"{what}{prefix}: mean {mean} vs {mean_ref}"
    return err


def psnr_np(a, b):
    d = (255.0 * (a.double() - b.double()))
    return float(20 * np.log10(255.0) - 10 * torch.log10((d ** 2).mean()))


# ---- backward fixtures (tests/golden/make_golden_bwd.py): per-parameter gradient summaries
BWD_FULL = ("embedder.unet.outc.weight", "embedder.unet.msg_processor.msg_embeddings.weight", "detector.pixel_decoder.linear.weight",
            "detector.pixel_decoder.linear.bias", "embedder.unet.inc.double_conv.1.weight")


def projection_vector(name: str, numel: int):
    """+-1 vector seeded by the parameter name (torch's CPU mt19937 stream for randint: stable across runs and platforms)"""
    import torch
    g = torch.Generator().manual_seed(sum(name.encode()) * 7919 + numel)
    return (torch.randint(0, 2, (numel,), generator=g, dtype=torch.int64) * 2 - 1).double()
