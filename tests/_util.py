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


# ---- thresholded bit decisions (north_star: "bit decisions bit-exact")
DECISION_MARGIN = 2e-5      # ~10 x the largest logit error of the HIP path measured on MI355X on identical inputs (1.5e-6)


def assert_decisions(preds, gold, margin=DECISION_MARGIN, what="", min_sure=0.999):
    """`preds > 0` equals `gold > 0` for every logit of the reference that lies further from 0 than `margin`; the logits inside the margin
    (where an fp32 re-ordering may legitimately flip the sign) must be a negligible share, so the mask cannot excuse a real difference.
    With VS_DECISION_LOG set, every call appends its measured error / smallest |logit| / masked count to that file (one GPU run then shows
    how far each margin is from the noise it excuses)."""
    import os
    preds, gold = preds.detach().cpu().float(), gold.detach().cpu().float()
    sure = gold.abs() > margin
    err = float((preds - gold).abs().max())
    flips = ((preds > 0) != (gold > 0))
    log = os.environ.get("VS_DECISION_LOG")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps(dict(what=what or os.environ.get("PYTEST_CURRENT_TEST", ""), margin=margin, err=err, numel=gold.numel(),
                                    min_abs=float(gold.abs().min()), masked=int((~sure).sum()), flips_all=int(flips.sum()),
                                    flips_sure=int((flips & sure).sum()))) + "\n")
    if os.environ.get("VS_DECISION_DISCOVER"):          # measuring run: record, do not judge
        return err
    assert int((~sure).sum()) <= max(1, int((1.0 - min_sure) * gold.numel())), f"{what}: {int((~sure).sum())} of {gold.numel()} reference logits inside the margin {margin}"
    assert not (flips & sure).any(), f"{what}: {int((flips & sure).sum())} bit decisions differ from the reference (max logit error {err:.2e})"
    return err
