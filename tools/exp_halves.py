#!/usr/bin/env python
"""Experiment: ONE 32-frame step as two 16-frame halves on two HIP streams (two model instances = separate workspaces), against the
single 32-frame step.  Does the second stream fill the drain / fill gaps between the dependent launches of the first?
usage: python tools/exp_halves.py [image|video] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import videoseal_amd
from bench import synthetic_batch

mode = sys.argv[1] if len(sys.argv) > 1 else "image"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
is_video = mode == "video"
B, S = 32, 768
frames = synthetic_batch(B, S, dev, seed=1000)
models = [videoseal_amd.build("videoseal_1.0", seed=0).eval().to(dev) for _ in range(2)]
for m in models:
    m.chunk_size = max(m.chunk_size, B)
cfg = models[0].embedder.cfg
gm = torch.Generator().manual_seed(5)
msgs = torch.randint(0, 2, (1 if is_video else B, cfg.nbits), generator=gm)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def whole():
    w = models[0].embed(frames, msgs, is_video=is_video)["imgs_w"]
    return models[0].detect(w, is_video=is_video)["preds"]


def halves():
    cur = torch.cuda.current_stream()
    outs = [None, None]
    for i, (m, s) in enumerate(zip(models, streams)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            fr = frames[16 * i:16 * i + 16]
            mg = msgs if is_video else msgs[16 * i:16 * i + 16]
            w = m.embed(fr, mg, is_video=is_video)["imgs_w"]
            outs[i] = m.detect(w, is_video=is_video)["preds"]
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs, 0) if not is_video else outs


def timeit(fn):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {}
for rep in range(3):
    res[f"whole_{rep}"] = round(timeit(whole), 3)
    res[f"halves_{rep}"] = round(timeit(halves), 3)
a, b = whole(), halves()
if not is_video:
    res["max_abs_logit_diff"] = float((a - b).abs().max())
print(json.dumps({"mode": mode, "ms_per_32_frames": res}))
