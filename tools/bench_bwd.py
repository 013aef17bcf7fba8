#!/usr/bin/env python
"""Time of the detector fine-tuning step (videoseal_amd.training.DetectorStep: forward that keeps its operands + loss + backward) on the
VideoSeal 1.0 extractor, against the inference forward of the same frames.  usage: tools/bench_bwd.py [batch]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import videoseal_amd
from videoseal_amd.training import DetectorStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = videoseal_amd.build("videoseal_1.0").eval().to("cuda")      # seeded random weights of the card's architecture (no checkpoint offline)
g = torch.Generator().manual_seed(1)
imgs = torch.rand(B, 3, 256, 256, generator=g).cuda()
msgs = torch.randint(0, 2, (B, model.embedder.cfg.nbits), generator=g)
step = DetectorStep(model)


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_inf = timed(lambda: model.detector(imgs), 5)
t_step = timed(lambda: step.step(imgs, msgs, accumulate=False), 3)
gmac = 5.82 * B        # 5.82 GMAC per frame forward; backward = 2x (data + weight products)
print(f"B={B} 256x256: inference forward {t_inf:.2f} ms; fine-tuning step (fwd + loss + bwd) {t_step:.2f} ms = "
      f"{3 * 2 * gmac / t_step:.1f} TFLOP/s of the 3 x forward MACs; {B / t_step * 1e3:.0f} frames/s")
