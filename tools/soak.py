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

# ---- the legacy card's extractor (matrix-core attention) and the training steps: every gradient tensor must repeat bit for bit
# (all reductions of the backward have a fixed order: no atomics anywhere)
from videoseal_amd.training import GeneratorStep

leg = videoseal_amd.build("videoseal_0.0", seed=0).eval().cuda()
x = torch.rand(16, 3, 256, 256, device="cuda")
ref = None
bad = 0
for it in range(N):
    p = leg.detect(x, is_video=True)["preds"]
    if ref is None:
        ref = p.clone()
    elif not torch.equal(p, ref):
        bad += 1
print(f"videoseal_0.0 detect (ViT, MFMA attention) B=16: {N} runs, {bad} differing", flush=True)
for card, B in (("videoseal_1.0", 8), ("videoseal_0.0", 4)):
    m = videoseal_amd.build(card, seed=0).cuda().train()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    frames = torch.rand(B, 3, 256, 256, device="cuda")
    masks = torch.ones(B, 1, 256, 256, device="cuda")
    msgs = torch.randint(0, 2, (B, m.embedder.cfg.nbits), generator=torch.Generator().manual_seed(5))
    gs = GeneratorStep(m, percep_loss="mse", percep_weight=1.0, decode_weight=1.0, balanced=True)
    ref = None
    bad = 0
    for it in range(max(3, N // 5)):
        m.load_state_dict(sd0)                      # (BatchNorm running statistics move with every training forward)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        gs.step(frames, masks, msgs)
        torch.cuda.synchronize()
        g = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        if ref is None:
            ref = g
        elif not all(torch.equal(g[k], ref[k]) for k in ref):
            bad += 1
    print(f"{card} generator step B={B}: {max(3, N // 5)} runs, {len(ref)} gradient tensors, {bad} runs differing", flush=True)
