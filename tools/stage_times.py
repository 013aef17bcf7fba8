#!/usr/bin/env python
"""Where an image-mode step goes, layer by layer: HIP-event time of every vs_conv_gemm launch of embed + detect (engine.time_all_convs),
summed per layer signature, next to the whole embed / detect times.  `python tools/stage_times.py [--batch 32] [--size 768] [--video]`"""
import argparse
import collections
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--video", action="store_true")
    ap.add_argument("--card", default="videoseal_1.0")
    a = ap.parse_args()
    import videoseal_amd
    dev = torch.device("cuda", 0)
    model = videoseal_amd.build(a.card, seed=0).eval().to(dev)
    model.chunk_size = max(model.chunk_size, a.batch)
    frames = synthetic_batch(a.batch, a.size, dev, seed=1)
    msgs = torch.randint(0, 2, (1 if a.video else a.batch, model.embedder.cfg.nbits), generator=torch.Generator().manual_seed(5))
    for _ in range(3):
        w = model.embed(frames, msgs, is_video=a.video)["imgs_w"]
        model.detect(w, is_video=True)
    eng = model._engine()
    eng.kernel_timers, eng.time_all_convs = [], True
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    K = 5
    tot = [0.0, 0.0]
    for _ in range(K):
        ev[0].record()
        w = model.embed(frames, msgs, is_video=a.video)["imgs_w"]
        ev[1].record()
        model.detect(w, is_video=True)
        ev[2].record()
        torch.cuda.synchronize()
        tot[0] += ev[0].elapsed_time(ev[1]); tot[1] += ev[1].elapsed_time(ev[2])
    agg = collections.OrderedDict()
    for name, e0, e1, fl in eng.kernel_timers:
        t = agg.setdefault(name, [0, 0.0, 0.0])
        t[0] += 1; t[1] += e0.elapsed_time(e1); t[2] += fl
    rows = [{"layer": k, "launches_per_step": v[0] / K, "ms_per_step": round(v[1] / K, 4), "us_per_launch": round(1e3 * v[1] / v[0], 1),
             "tf_eq": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in agg.items()]
    print(json.dumps({"embed_ms": round(tot[0] / K, 3), "detect_ms": round(tot[1] / K, 3), "conv_ms": round(sum(r["ms_per_step"] for r in rows), 3),
                      "note": "event-bracketed launches include launch gaps; conv launches only (norm / shell / gather kernels are the remainder)",
                      "layers": rows}, indent=1))


if __name__ == "__main__":
    main()
