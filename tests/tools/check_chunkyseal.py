#!/usr/bin/env python
"""ChunkySeal at its released size (dims 362/724/1448/2896, 774 M extractor parameters): HIP detect vs the CPU oracle on one frame.
Too heavy for the test suite (7 GB of synthetic weights, ~1.2 TFLOP per frame on the CPU); run by hand on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import videoseal_ref as R
from oracle.inputs import synthetic_frames
from oracle.weights import make_state_dict, spec_from_card
from tests.test_gpu_e2e import make_model

t0 = time.time()
spec = spec_from_card(os.path.join(os.path.dirname(__file__), "..", "..", "videoseal_amd", "cards", "chunkyseal.yaml"))
sd = make_state_dict(spec, seed=2)
print(f"state_dict: {sum(v.numel() for v in sd.values())/1e9:.2f} G values in {time.time()-t0:.0f} s", flush=True)
model = make_model(spec, sd)
imgs = synthetic_frames(2, 512, 480, seed=7)
t0 = time.time()
torch.set_num_threads(min(os.cpu_count() or 1, 64))
ref = R.detect(sd, spec, imgs)["preds"]
print(f"oracle detect: {time.time()-t0:.0f} s", flush=True)
got = model.detect(imgs.cuda(), is_video=True)["preds"].cpu()
err = (got - ref).abs().max().item()
flips = int(((got > 0) != (ref > 0)).sum())
print(f"chunkyseal detect 2 frames: max|logit - oracle| = {err:.3e} (|ref| max {ref.abs().max().item():.2f}), decisions flipped {flips} / {ref.numel()}")
