#!/usr/bin/env python
"""Numerical diagnostics on the GPU box: error magnitudes of the HIP path vs the CPU oracle (not a test)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import videoseal_ref as R  # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs  # noqa: E402
from oracle.weights import make_state_dict, spec_from_card  # noqa: E402
from tests.test_gpu_e2e import make_model  # noqa: E402

spec = spec_from_card(os.path.join(ROOT, "videoseal_amd", "cards", "videoseal_1.0.yaml"))
sd = make_state_dict(spec, seed=0)
model = make_model(spec, sd)
tot_flip = tot = 0
for seed, (n, h, w) in enumerate([(8, 768, 768), (8, 256, 256), (6, 480, 640), (8, 768, 768)]):
    imgs = synthetic_frames(n, h, w, seed=100 + seed, kind="smooth" if seed < 3 else "uniform")
    msgs = synthetic_msgs(1, spec.nbits, seed=seed)
    model.chunk_size, model.step_size = 32, 4
    out = model.embed(imgs.cuda(), msgs, is_video=True)
    ref = R.embed_video(sd, spec, imgs, msgs)
    e_img = (out["imgs_w"].cpu() - ref["imgs_w"]).abs().max().item()
    p = model.detect(ref["imgs_w"].cuda(), is_video=True)["preds"].cpu()
    pr = R.detect(sd, spec, ref["imgs_w"])["preds"]
    e_log = (p - pr).abs().max().item()
    flips = ((p > 0) != (pr > 0)).sum().item()
    tot_flip += flips; tot += p.numel()
    print(f"case {seed} {n}x{h}x{w}: max|imgs_w err|={e_img:.2e} psnr(ours)={R.psnr(out['imgs_w'].cpu(), imgs, True):.4f} "
          f"psnr(ref)={R.psnr(ref['imgs_w'], imgs, True):.4f} max|logit err|={e_log:.2e} mean|logit err|={(p-pr).abs().mean():.2e} "
          f"min|logit|={pr.abs().min():.2e} flips={flips}/{p.numel()}")
print(f"total decision flips on identical inputs: {tot_flip}/{tot}")
