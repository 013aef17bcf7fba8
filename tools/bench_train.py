#!/usr/bin/env python
"""The generator-side training step (videoseal_amd.training.GeneratorStep: differentiable forward + decoding / yuv loss + backward of embedder
AND extractor) on VideoSeal 1.0, 16 frames of 256x256: wall time per step, host time per step (how long the Python side needs to issue it)
and, with --torch-profile, the ATen / HIP kernel table.  usage: tools/bench_train.py [--torch-profile]   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
r = bench.gen_step_leg(dev)
print(r)
if "--torch-profile" in sys.argv:
    import videoseal_amd
    from torch.profiler import profile, ProfilerActivity
    from videoseal_amd.training import GeneratorStep
    model = videoseal_amd.build("videoseal_1.0", seed=0).to(dev).train()
    frames = bench.synthetic_batch(16, 256, dev, seed=7)
    masks = torch.ones(16, 1, 256, 256, device=dev)
    msgs = torch.randint(0, 2, (16, 256), generator=torch.Generator().manual_seed(5))
    gs = GeneratorStep(model, percep_loss="yuv", percep_weight=0.1, decode_weight=1.0, balanced=False)
    for _ in range(2):
        model.zero_grad(set_to_none=True); gs.step(frames, masks, msgs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.zero_grad(set_to_none=True); gs.step(frames, masks, msgs)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host time to issue one step: {t_host * 1e3:.1f} ms (wall incl. GPU: {(time.perf_counter() - t0) * 1e3:.1f} ms)")
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model.zero_grad(set_to_none=True); gs.step(frames, masks, msgs)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
