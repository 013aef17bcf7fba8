#!/usr/bin/env python
"""Race hunt: the same embed + detect call repeated N times must return bit-identical results every time (the producer/consumer
kernels synchronise with hand-placed barriers and counted waits; a missing one shows up as run-to-run differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import videoseal_amd
from videoseal_amd.capi import CModel

N = int(os.environ.get("N", 25))
model = videoseal_amd.build("videoseal_1.0", seed=0).eval().cuda()
for B, S, video, lowres in ((32, 768, False, False), (32, 768, True, False), (16, 768, True, True), (3, 250, False, False), (7, 130, True, True)):
    model.chunk_size = max(model.chunk_size, B)
    x = torch.rand(B, 3, S, S + (S % 7), device="cuda")
    msgs = torch.randint(0, 2, (1 if video else B, 256))
    ref_w = ref_p = None
    bad = 0
    for it in range(N):
        w = model.embed(x, msgs, is_video=video, lowres_attenuation=lowres)["imgs_w"]
        p = model.detect(w, is_video=True)["preds"]
        if ref_w is None:
            ref_w, ref_p = w.clone(), p.clone()
        elif not (torch.equal(w, ref_w) and torch.equal(p, ref_p)):
            bad += 1
    print(f"python host  B={B} S={S} video={video} lowres={lowres}: {N} runs, {bad} differing", flush=True)
    cm = CModel(model.embedder.cfg, model.state_dict())
    ref_w = ref_p = None
    bad = 0
    for it in range(N):
        w = cm.embed(x, msgs, step=(4 if video else 1), lowres_attenuation=lowres)
        p = cm.detect(w)
        if ref_w is None:
            ref_w, ref_p = w.clone(), p.clone()
        elif not (torch.equal(w, ref_w) and torch.equal(p, ref_p)):
            bad += 1
    print(f"model C-ABI  B={B} S={S} video={video} lowres={lowres}: {N} runs, {bad} differing", flush=True)
    del cm
