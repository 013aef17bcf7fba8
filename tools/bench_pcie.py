#!/usr/bin/env python
"""PCIe-inclusive rates of the embed + extract path when the caller's frames live in HOST memory (the reference's own calling convention:
videoseal/models/videoseal.py:151-199 and inference_streaming.py:83-164 read frames on the CPU, move them chunk by chunk and bring the
watermarked frames back).  bench.py's `value` is quoted with the inputs resident in HBM; this tool states what the same configs[1] workload
(32 frames of 768 x 768, 256 bits) reaches when every frame crosses the link twice:

  resident         inputs and outputs stay in HBM (= bench.py)
  host_pageable    CPU fp32 tensor straight into model.embed / model.detect, results returned to the CPU (the drop-in call, nothing else)
  host_pinned      the caller's buffers are pinned, explicit non-blocking copies, one stream (copy - compute - copy in sequence)
  host_pinned_2buf pinned + double-buffered: H2D of batch i+1 and D2H of batch i-1 on their own HIP streams under the compute of batch i
  u8_pinned_2buf   uint8 RGB24 frames in and out (embed_u8 / detect_u8, the data format of inference_streaming.py), 4 x fewer bytes

usage: tools/bench_pcie.py [--batches 8] [--batch 32] [--size 768]      (GPU box)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import videoseal_amd

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=768)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, S, n = args.batch, args.size, args.batches
model = videoseal_amd.build("videoseal_1.0", seed=0).eval().to(dev)
nbits = model.embedder.cfg.nbits
msgs = torch.randint(0, 2, (B, nbits), generator=torch.Generator().manual_seed(5))
msgs_v = msgs[:1]
frames_dev = bench.synthetic_batch(B, S, dev, seed=1000)
frames_cpu = frames_dev.cpu()
u8_dev = (frames_dev * 255.0).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
model.chunk_size = max(model.chunk_size, B)
cur = torch.cuda.current_stream()


def timed(fn, warm=2):
    for _ in range(warm):
        fn(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames_per_s": round(n * B / dt, 1), "ms_per_batch": round(dt / n * 1e3, 2)}


def resident(k):
    for _ in range(k):
        w = model.embed(frames_dev, msgs, is_video=False)["imgs_w"]
        model.detect(w, is_video=True)["preds"]


def host_pageable(k):
    for _ in range(k):
        w = model.embed(frames_cpu, msgs, is_video=False)["imgs_w"]          # comes back on the CPU, as the reference returns it
        model.detect(w, is_video=True)["preds"]


pin_in = frames_cpu.pin_memory()
pin_out = [torch.empty_like(pin_in).pin_memory() for _ in range(2)]
pin_preds = [torch.empty(B, 1 + nbits).pin_memory() for _ in range(2)]


def host_pinned(k):
    for i in range(k):
        x = pin_in.to(dev, non_blocking=True)
        w = model.embed(x, msgs, is_video=False)["imgs_w"]
        p = model.detect(w, is_video=True)["preds"]
        pin_out[i & 1].copy_(w, non_blocking=True)
        pin_preds[i & 1].copy_(p, non_blocking=True)


s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def two_buffers(k, src, dst, dev_in, emb, det):
    """H2D on s_in, compute on the current stream, D2H on s_out; two device input buffers and two pinned output buffers in rotation"""
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_free = [None, None]            # compute that last read dev_in[b]
    ev_out = [None, None]             # D2H that last wrote dst[b]

    def prefetch(i):
        b = i & 1
        with torch.cuda.stream(s_in):
            if ev_free[b] is not None:
                s_in.wait_event(ev_free[b])
            dev_in[b].copy_(src, non_blocking=True)
            ev_in[b].record(s_in)

    prefetch(0)
    for i in range(k):
        b = i & 1
        cur.wait_event(ev_in[b])
        if i + 1 < k:
            prefetch(i + 1)
        w = emb(dev_in[b])
        p = det(w)
        e = torch.cuda.Event(); e.record(cur); ev_free[b] = e
        if ev_out[b] is not None:
            ev_out[b].synchronize()                    # the host buffer is about to be overwritten (a real caller has consumed it by now)
        with torch.cuda.stream(s_out):
            s_out.wait_event(e)
            dst[b].copy_(w, non_blocking=True)
            pin_preds[b].copy_(p, non_blocking=True)
            w.record_stream(s_out); p.record_stream(s_out)
            eo = torch.cuda.Event(); eo.record(s_out); ev_out[b] = eo
    cur.wait_stream(s_out)


dev_in_f = [torch.empty_like(frames_dev) for _ in range(2)]


def host_pinned_2buf(k):
    two_buffers(k, pin_in, pin_out, dev_in_f, lambda x: model.embed(x, msgs, is_video=False)["imgs_w"],
                lambda w: model.detect(w, is_video=True)["preds"])


u8_pin_in = u8_dev.cpu().pin_memory()
u8_pin_out = [torch.empty_like(u8_pin_in).pin_memory() for _ in range(2)]
dev_in_u8 = [torch.empty_like(u8_dev) for _ in range(2)]


def u8_resident(k):
    for _ in range(k):
        w = model.embed_u8(u8_dev, msgs_v)["imgs_w"]
        model.detect_u8(w)["preds"]


def u8_pinned_2buf(k):
    two_buffers(k, u8_pin_in, u8_pin_out, dev_in_u8, lambda x: model.embed_u8(x, msgs_v)["imgs_w"], lambda w: model.detect_u8(w)["preds"])


def link_rate():
    """what the link itself does with these buffers (GB/s, one direction at a time)"""
    x = torch.empty_like(frames_dev)
    out = {}
    for name, fn in (("h2d_pinned", lambda: x.copy_(pin_in, non_blocking=True)), ("d2h_pinned", lambda: pin_out[0].copy_(x, non_blocking=True)),
                     ("h2d_pageable", lambda: x.copy_(frames_cpu)), ("d2h_pageable", lambda: frames_cpu.copy_(x))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        out[name + "_GBps"] = round(4 * x.numel() * 4 / (time.perf_counter() - t0) / 1e9, 1)
    return out


res = {"workload": f"videoseal_1.0 256-bit, {B} frames {S}x{S}, embed + extract, {n} batches per measurement",
       "bytes_per_batch_each_way_MB": {"fp32": round(frames_dev.numel() * 4 / 1e6, 1), "u8": round(u8_dev.numel() / 1e6, 1)},
       "link": link_rate(),
       "image mode, fp32 frames": {"resident": timed(resident), "host_pageable": timed(host_pageable), "host_pinned": timed(host_pinned),
                                   "host_pinned_2buf": timed(host_pinned_2buf)},
       "video mode (step 4), uint8 RGB24 frames": {"resident": timed(u8_resident), "u8_pinned_2buf": timed(u8_pinned_2buf)}}
print(json.dumps(res, indent=1))
