#!/usr/bin/env python
"""bench.py -- frames/sec of VideoSeal 1.0 embed + extract on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode image|video] [--batch 32] [--size 768]

A "step" = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
    model.embed(batch)  ->  model.detect(watermarked batch)  [-> all-gather of the bit logits when N > 1]
Workload at N=1 = BASELINE.json configs[1]: VideoSeal 1.0, 256 bits, 32 frames of 768x768, image mode
(embedder on every frame, full-resolution JND), random-init weights of the card's architecture.
N > 1: one process per GPU (torchrun), the same batch per GPU (weak scaling), frames sharded over ranks,
no data-path collective for embedding and one RCCL all-gather of the [32, 257] logits per step for extraction.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     -- dominant kernel = the U-Net bottleneck 3x3 conv (384->384 @32x32, 81 % of the embed FLOPs):
                  algorithmic FLOPs per launch / average launch duration measured live with HIP events on the
                  launch stream, against 2500/3 = 833.3 TFLOP/s (dense f16 MFMA peak / 3 partial products of the default
                  2 x f16 operand split; 2500/6 with VIDEOSEAL_CONV=bf16x3); `frac` from the live events, `frac_rocprof` from the tracked
                  rocprofv3 kernel-trace average of the same command (profiles/), `e2e_frac` = whole-step model
                  FLOP/s over the same ceiling; `shell` = the HBM-bound kernels against 8 TB/s
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference path, pinned to the reference by
                  tests/golden) timed on this host's cores on a bounded sample of the same workload
--mode chain = BASELINE configs[2]: 16-frame 768x768 clip -> embed -> JPEG/Crop/Resize/Brightness/Contrast/
Saturation/Hue at the fixed validation strengths -> detect (the north-star "fused embed -> augment -> extract").
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.time()      # (before `import torch`: the first import on a fresh box pages the image in)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA
# split arithmetic: every fp32 product = 3 f16 MFMA partial products (2 x f16 split, the default) or 6 bf16 ones (3 x bf16 exact
# split), fp32 accumulate; f16 and bf16 MFMA have the same dense peak, so the ceiling for *algorithmic* (fp32-equivalent) FLOP/s
# is that peak / 3 resp. / 6
def products(eng):
    return 3 if eng.arith == 2 else 6


def peak_split(eng):
    return PEAK_BF16_MFMA_TFLOPS / products(eng)


def arith_name(eng):
    return "2 x f16 operand split, 3 products on v_mfma_f32_32x32x16_f16" if eng.arith == 2 else "3 x bf16 exact operand split, 6 products on v_mfma_f32_32x32x16_bf16"


def current_profile(key):
    """path of the tracked profile summary `key` named by profiles/CURRENT.json (None if absent) -- an explicit pointer, not 'newest by name'"""
    idx = os.path.join(ROOT, "profiles", "CURRENT.json")
    if not os.path.exists(idx):
        return None
    name = json.load(open(idx)).get(key)
    path = os.path.join(ROOT, "profiles", name) if name else None
    return path if path and os.path.exists(path) else None


def arith_used(eng):
    """which operand split each network actually ran on (the range guard may have switched one to 3 x bf16: half the matrix rate)"""
    if not eng.use_split:
        return {"embedder": "f32", "extractor": "f32"}
    nm = {2: "f16x2", 3: "bf16x3"}
    return {"embedder": nm[eng.arith_net["E"]], "extractor": nm[eng.arith_net["X"]], "extractor_layers_pinned_to_bf16x3": len(eng.layer_arith),
            "selection": "auto (range guard, per layer for the ConvNeXt extractor)" if eng.auto_arith else "forced"}


def synthetic_batch(n, size, device, seed):
    """device-resident synthetic frames in [0,1]: low-pass noise + noise (content does not change the work)."""
    g = torch.Generator(device=device).manual_seed(seed)
    lo = torch.rand(n, 3, size // 16, size // 16, device=device, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(size, size), mode="bilinear", align_corners=False)
    x = 0.85 * x + 0.15 * torch.rand(n, 3, size, size, device=device, generator=g)
    return x.clamp_(0, 1).contiguous()


CHAIN_ARGS = (40, 0.71, 0.71, 0.5, 1.5, 1.5, 0.1)     # JPEG q, Crop, Resize, Brightness, Contrast, Saturation, Hue (augmentation/__init__.py:107-123)


def cpu_baseline(card_path, size, mode, step_size, max_seconds=25.0):
    """Oracle on the host cores: frames/sec on a bounded sample of the same workload (8 frames, 1 warm-up, >= 2 reps per thread count).
    torch's CPU convolutions stop scaling beyond a few dozen threads (measured: see below), so the sample is timed with 32 threads;
    `cores` is the thread count that produced the reported value."""
    from oracle import augment as A
    from oracle import videoseal_ref as R
    from oracle.inputs import synthetic_frames, synthetic_msgs
    from oracle.weights import make_state_dict, spec_from_card
    spec = spec_from_card(card_path)
    sd = make_state_dict(spec, seed=0)
    n = 8
    imgs = synthetic_frames(n, size, size, seed=0, kind="uniform")

    def run():
        with torch.no_grad():
            if mode == "image":
                out = R.embed_image(sd, spec, imgs, synthetic_msgs(n, spec.nbits))
            else:
                out = R.embed_video(sd, spec, imgs, synthetic_msgs(1, spec.nbits), step_size=step_size)
            w = out["imgs_w"]
            if mode == "chain":
                q, cs, rs, br, co, sa, hu = CHAIN_ARGS
                w = A.jpeg(w, q)
                th, tw = int(cs * size), int(cs * size)
                w = A.crop(w, (size - th) // 2, (size - tw) // 2, th, tw)
                w = A.resize(w, (int(rs * th), int(rs * tw)))
                w = A.hue(A.saturation(A.contrast(A.brightness(w, br), co), sa), hu)
            R.detect(sd, spec, w)

    # the UNMODIFIED reference module instead of the port whenever a checkout is reachable (VIDEOSEAL_REFERENCE_ROOT, or /root/reference in the
    # build container; the GPU box has neither -> "port").  Same state_dict, same sample; the chain's augmentations need torchvision -> port.
    kind, ref_model = "port", None
    ref_root = os.environ.get("VIDEOSEAL_REFERENCE_ROOT") or ("/root/reference" if os.path.isdir("/root/reference/videoseal") else None)
    if ref_root and mode in ("image", "video") and os.environ.get("VS_BENCH_CPU_PORT") != "1":
        try:
            import yaml
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            import make_golden as MG                    # the stub-import recipe of SURVEY appendix A (timm / torchvision / cv2 / av stand-ins)
            MG.REF = ref_root
            ref_model = MG.build_reference(spec, yaml.safe_load(open(os.path.join(ref_root, "videoseal", "cards", os.path.basename(card_path))))).eval()
            ref_model.load_state_dict(sd, strict=True)
            ref_model.step_size = step_size
            kind = "reference"
        except Exception as e:                          # a checkout that does not import here: say so and time the port
            print(f"cpu_baseline: reference at {ref_root} not usable ({e!r}); timing the port", file=sys.stderr)
            ref_model = None
    if ref_model is not None:
        def run():              # noqa: F811
            with torch.no_grad():
                if mode == "image":
                    w = ref_model.embed(imgs, synthetic_msgs(n, spec.nbits), is_video=False)["imgs_w"]
                else:
                    w = ref_model.embed(imgs, synthetic_msgs(1, spec.nbits), is_video=True)["imgs_w"]
                ref_model.detect(w, is_video=(mode != "image"))

    host = os.cpu_count() or 1
    tried = {}
    # every host core is opt-in (VS_BENCH_CPU_ALL_CORES=1): on the 256-CPU GPU box 256 threads ran the same sample 83x SLOWER than
    # 32 threads (0.086 vs 7.18 frames/s, profiles/archive/r02a_cpu_threads.json) and took 5 minutes of the run
    for cores in sorted({min(host, 32)} | ({host} if os.environ.get("VS_BENCH_CPU_ALL_CORES") == "1" else set())):
        torch.set_num_threads(cores)
        t0 = time.time(); run(); warm = time.time() - t0
        reps, t_used, times = 0, 0.0, []
        while reps < 2 or (t_used + warm + min(times) < max_seconds / 2 and reps < 4):
            t0 = time.time(); run(); dt = time.time() - t0
            times.append(dt); t_used += dt; reps += 1
        tried[cores] = round(n / min(times), 3)
    cores = max(tried, key=tried.get)
    cal = current_profile("cpu_port_vs_reference")
    ratio = json.load(open(cal)) if cal else None
    return {"value": tried[cores], "port_over_reference": (ratio["port_over_reference"] if ratio else None),
            "port_over_reference_note": (ratio["note"] if ratio else None), "unit": "frames/s", "cores": cores, "host_cpus": host, "kind": kind,
            "by_threads": {str(k): v for k, v in tried.items()},
            "sample": f"{n} frames {size}x{size}, {mode} mode, embed" + ("+augment chain" if mode == "chain" else "") + "+detect, best rep after 1 warm-up per thread count, "
                      + ("the UNMODIFIED reference module (checkout at %s, stub-import recipe of tests/golden/make_golden.py), seeded state_dict" % ref_root
                         if kind == "reference" else
                         "torch fp32 CPU oracle (restatement of the reference path pinned by tests/golden; the reference package itself "
                         "is not present on the GPU box)")}


def measure_sustained_mfma():
    """bf16 dense TFLOP/s of a register-only MFMA loop with random operands on this GPU (None if the micro-benchmark is not built)."""
    import ctypes
    so = os.path.join(ROOT, "tools", "micro", "libmfma_peak.so")
    if not os.path.exists(so):
        return None
    lib = ctypes.CDLL(so)
    out = torch.zeros(1024, device="cuda")
    ms = ctypes.c_float()
    best = 0.0
    for _ in range(3):
        blocks, threads, iters = 1024, 512, 1000
        if lib.run_mfma(0, blocks, threads, iters, 0, 1, ctypes.c_void_p(out.data_ptr()), ctypes.byref(ms)) != 0:
            return None
        best = max(best, blocks * threads // 64 * iters * 24 * 32 * 32 * 16 * 2 / ms.value / 1e9)
    return best


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: 30 timed steps -- the barrier + synchronize pair that brackets the timed region costs ~0.9 ms, 1.2 % of ten 7.6 ms steps and 0.4 % of thirty:
    #  profiles/r06zj_bench_steps_sweep.txt)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["image", "video", "stream", "chain"], default="image",
                    help="stream = BASELINE config 4: --frames frames processed as 16-frame embed(lowres_attenuation)+detect calls; "
                         "chain = BASELINE config 3: 16-frame clip -> embed -> JPEG/Crop/Resize/colour chain -> detect")
    ap.add_argument("--capi", action="store_true", help="drive the model-level C-ABI (vs_model_embed / vs_model_detect, host code in C++, static "
                    "tile heuristics) instead of the Python host path")
    ap.add_argument("--pipeline", action="store_true", help="image / video mode: overlap detect(batch i) with embed(batch i+1) on two HIP streams")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="stream mode: strictly sequential calls on one stream "
                    "(default: detect(chunk i) on a second HIP stream while embed(chunk i+1) is issued, videoseal_amd/streaming.py)")
    ap.add_argument("--u8", action="store_true", help="stream mode: uint8 RGB24 clips in and out (inference_streaming.py's data format) "
                    "through embed_u8 / detect_u8 instead of fp32 NCHW tensors")
    ap.add_argument("--frames", type=int, default=1024, help="stream mode: total frames of the clip (sharded over the ranks)")
    ap.add_argument("--group", type=int, default=None, help="stream mode: 16-frame chunks whose key frames share one U-Net pass (default: enough "
                    "for 32 key frames = 8 chunks; 1 = the literal per-chunk calls of round 3)")
    ap.add_argument("--det-batch", type=int, default=None, help="stream mode: frames per extractor pass (default 32)")
    ap.add_argument("--graphs", action="store_true", help="replay the per-chunk launch sequences from hipGraphs")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU and step (default 32; 16 in chain mode)")
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--card", default="videoseal_1.0")
    ap.add_argument("--conv", choices=["auto", "f16x2", "bf16x3", "f32"], default=None, help="arithmetic of the dense layers (VIDEOSEAL_CONV; default auto = "
                    "2 x f16 split behind the range guard, which falls back to the exact 3 x bf16 split per network)")
    ap.add_argument("--lowres-attenuation", action="store_true")
    ap.add_argument("--detect-only", action="store_true", help="time model.detect() only (BASELINE config 5: ChunkySeal extractor)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="time only the CPU leg (no GPU needed) and print its JSON object")
    ap.add_argument("--dump-preds", default=None, help="rank 0 saves the (gathered) logits of the last timed step to this path (torch.save)")
    ap.add_argument("--no-extra", action="store_true", help="default run only: skip the short legs over the other BASELINE configs "
                    "(video mode, configs[2] chain, configs[3] streaming, configs[4] ChunkySeal detect, the training step)")
    return ap.parse_args(argv)



def coll_device(dev):
    """where the scalar collectives of this file (rank count, max-over-ranks time) live: the GPU under RCCL, the host under a gloo group"""
    return torch.device("cpu") if torch.distributed.get_backend() == "gloo" else dev


def run(args):
    """one workload: returns the JSON line (rank 0) or None"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("VS_BENCH_FORCE_DIST") == "1"   # the env switch exercises the RCCL path on one GPU
    if args.gpus != world and dist_on:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and not dist_on:       # (main() re-executes a plain `python bench.py --gpus N` under the launcher before it gets here)
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # VS_BENCH_COLLECTIVE=gloo: the ranks exchange through a host-side process group (logits staged in pinned memory, dist.py) and may SHARE a
    # device (rank r -> cuda:(r mod visible devices)): RCCL refuses two ranks on one GPU, and a 1-GPU box is all the build has -- this is how the
    # multi-rank code path of this file (sharding, gather, max-over-ranks timing, one line from rank 0) is executed before an 8-GPU node runs it
    host_coll = os.environ.get("VS_BENCH_COLLECTIVE", "nccl") == "gloo"
    if host_coll:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on and not torch.distributed.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if host_coll:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=dev)
    n_ranks_seen = None
    if dist_on:      # how many ranks the communicator itself sees: a sum of ones over it (not the launcher's environment variable)
        one = torch.ones(1, device=coll_device(dev), dtype=torch.int32)
        torch.distributed.all_reduce(one)
        n_ranks_seen = int(one.item())

    if args.graphs:
        os.environ["VIDEOSEAL_GRAPHS"] = "1"
    conv_env = os.environ.get("VIDEOSEAL_CONV")
    if args.conv:
        os.environ["VIDEOSEAL_CONV"] = args.conv          # read when the engine is built (first model call)
    try:
        return _run(args, world, rank, dev, dist_on, n_ranks_seen)
    finally:
        if args.conv:
            if conv_env is None:
                os.environ.pop("VIDEOSEAL_CONV", None)
            else:
                os.environ["VIDEOSEAL_CONV"] = conv_env


def _run(args, world, rank, dev, dist_on, n_ranks_seen):
    import videoseal_amd
    from videoseal_amd.dist import gather_frame_logits, shard_range
    model = videoseal_amd.build(args.card, seed=0).eval().to(dev)
    cfg = model.embedder.cfg
    chain = args.mode == "chain"
    if args.batch is None:
        args.batch = 16 if chain else 32
    B, S = args.batch, args.size
    stream = args.mode == "stream"
    if stream:       # strong scaling: the clip is split into contiguous 16-aligned frame ranges
        f0, f1 = shard_range(args.frames, rank, world, 16)
        B = f1 - f0
    frames = synthetic_batch(B, S, dev, seed=1000 + rank)
    frames_u8 = (frames * 255.0).to(torch.uint8).permute(0, 2, 3, 1).contiguous() if (stream and args.u8) else None
    gm = torch.Generator().manual_seed(5)
    is_video = args.mode in ("video", "stream", "chain")
    msgs = torch.randint(0, 2, (1 if is_video else B, cfg.nbits), generator=gm)
    model.chunk_size = max(model.chunk_size, B)

    two_streams = stream and args.overlap

    def step_stream_overlapped():     # videoseal_amd/streaming.py: detect(chunk i) on a second HIP stream while embed(chunk i+1) is issued
        from videoseal_amd.streaming import embed_detect_chunks
        preds = embed_detect_chunks(model, frames_u8 if args.u8 else frames, msgs, chunk=16, lowres_attenuation=True, overlap=True, group=args.group, det_batch=args.det_batch)
        if dist_on:
            preds = gather_frame_logits(preds, args.frames, align=16)
        return preds

    def step_stream():      # inference_streaming.py:83-107,117-164: 16-frame chunks, low-res attenuation, mean of the logits
        if two_streams:
            return step_stream_overlapped()
        logits = []
        for a in range(0, B, 16):
            if args.u8:
                w = model.embed_u8(frames_u8[a:a + 16], msgs, lowres_attenuation=True)["imgs_w"]
                logits.append(model.detect_u8(w)["preds"])
            else:
                w = model.embed(frames[a:a + 16], msgs, is_video=True, lowres_attenuation=True)["imgs_w"]
                logits.append(model.detect(w, is_video=True)["preds"])
        preds = torch.cat(logits, 0)
        if dist_on:
            preds = gather_frame_logits(preds, args.frames, align=16)
        return preds

    cmodel = None
    if args.capi:
        from videoseal_amd.capi import CModel
        cmodel = CModel(cfg, model.state_dict(), scaling_w=model.blender.scaling_w, scaling_i=model.blender.scaling_i)

    aug_timers = []

    chain_seq = [None]

    def step_chain():
        """BASELINE configs[2]: clip -> embed (key frames every step_size) -> the fixed-strength validation chain -> detect.  The chain goes through
        augmentation.Sequential as the reference's combined validation augmentations do (augmentation/__init__.py:107-123, sequential.py:8-30);
        round 5: Sequential runs Crop + Resize + Brightness as one kernel and Contrast + Saturation + Hue as one pass (bit-identical values)."""
        from videoseal_amd import augmentation as G
        if chain_seq[0] is None:
            chain_seq[0] = G.Sequential(G.JPEG(), G.Crop(), G.Resize(), G.Brightness(), G.Contrast(), G.Saturation(), G.Hue())
            chain_seq.append(G.Sequential(*chain_seq[0].transforms[1:]))      # the same chain behind a separately timed JPEG round trip
        # Crop.plan() draws its window with torch.randint (geometric.py:128-150): the same window every step, so that --dump-preds and the
        # per-kernel timers see one workload from run to run
        torch.manual_seed(1234)
        w = model.embed(frames, msgs, is_video=True, lowres_attenuation=args.lowres_attenuation)["imgs_w"]
        timed = eng_ref[0] is not None and eng_ref[0].shell_timers is not None
        G.TIMERS = aug_timers if timed else None
        try:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                x = G.jpeg_compress(w, CHAIN_ARGS[0])
                e1.record()
                aug_timers.append(("aug:jpeg_roundtrip", e0, e1, 2 * w.numel() * 4))
                x, _ = chain_seq[1](x, None, CHAIN_ARGS[1:])
            else:
                x, _ = chain_seq[0](w, None, CHAIN_ARGS)
        finally:
            G.TIMERS = None
        return model.detect(x, is_video=True)["preds"]

    eng_ref = [None]

    def step():
        if stream:
            return step_stream()
        if chain:
            preds = step_chain()
            if dist_on:
                preds = gather_frame_logits(preds, B * world, align=B)
            return preds
        if cmodel is not None:
            w = cmodel.embed(frames, msgs, step=(cfg.step_size if is_video else 1), lowres_attenuation=args.lowres_attenuation)
            return cmodel.detect(w)
        if args.detect_only:
            preds = model.detect(frames, is_video=True)["preds"]
        elif args.pipeline and not dist_on:
            # consecutive steps are independent batches: detect(batch i) runs on a second HIP stream while embed(batch i+1) is issued
            # (same calls, same results: videoseal_amd/streaming.py).  The timed region ends with a device-wide synchronize.
            from videoseal_amd.streaming import _streams
            s_emb, s_det = _streams(dev)
            s_emb.wait_stream(torch.cuda.current_stream()); s_det.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_emb):
                w = model.embed(frames, msgs, is_video=is_video, lowres_attenuation=args.lowres_attenuation)["imgs_w"]
                ev = torch.cuda.Event(); ev.record(s_emb)
            with torch.cuda.stream(s_det):
                s_det.wait_event(ev); w.record_stream(s_det)
                preds = model.detect(w, is_video=True)["preds"]
        else:
            out = model.embed(frames, msgs, is_video=is_video, lowres_attenuation=args.lowres_attenuation)
            preds = model.detect(out["imgs_w"], is_video=True)["preds"]
        if dist_on:
            preds = gather_frame_logits(preds, B * world, align=B)
        return preds

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng = model._engine()
    eng_ref[0] = eng
    if not args.no_kernel_timers:
        eng.kernel_timers = []
        eng.shell_timers = []
        eng.prof_extractor = bool(args.detect_only)
    barrier()
    t_ready = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds = step()
    host_issue = time.perf_counter() - t0            # how long the host needed to ISSUE the steps (the GPU runs behind it)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], device=coll_device(dev), dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same K steps once more with detect(batch i) overlapped with embed(batch i + 1) on a second HIP stream (what a serving loop does
    # with independent batches); reported next to `value`, which stays the sequential number the kernel timers belong to
    pipelined = None
    if (not args.pipeline and not dist_on and not stream and not chain and cmodel is None and not args.detect_only and not args.graphs
            and not args.no_kernel_timers):       # (profiled runs keep to the sequential steps the kernel statistics are quoted for)
        kt, sh = eng.kernel_timers, eng.shell_timers
        eng.kernel_timers = eng.shell_timers = None
        args.pipeline = True
        step(); barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        pipelined = B * args.steps / (time.perf_counter() - t1)
        args.pipeline = False
        eng.kernel_timers, eng.shell_timers = kt, sh

    roof = None
    if eng.kernel_timers:
        dur = [a.elapsed_time(b) * 1e-3 for _, a, b, _ in eng.kernel_timers]
        flops = eng.kernel_timers[0][3]
        avg = sum(dur) / len(dur)
        ach = flops / avg / 1e12
        split = eng.use_split
        nprod = products(eng)
        peak = peak_split(eng) if split else PEAK_F32_MFMA_TFLOPS
        prof_name = eng.kernel_timers[0][0]
        dom_is_bottleneck = prof_name.startswith("bott.")
        if dom_is_bottleneck:
            res = cfg.img_size // 2 ** (len(cfg.mults) - 1)           # (videoseal_1.0: 384 channels @32x32; the 0.0 card has its own width)
            kname = (("conv3x3_pl_kernel" if getattr(eng, "planes_chain_ran", False) else "conv3x3_patch_pc_kernel")
                     if split else "conv_gemm_kernel") + f" (U-Net bottleneck 3x3 conv {cfg.bott}->{cfg.bott} @{res}x{res}, "
        else:       # detect-only workloads: the extractor GEMM that carries most of the step (engine.prof_extractor names it)
            kname = prof_name + " ("
        roof = {"bound": "mfma",
                "kernel": kname + (arith_name(eng) + ", fp32 accumulate)" if split else "v_mfma_f32_32x32x2_f32)"),
                "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "peak_note": (f"2500 TF dense 16-bit MFMA / {nprod} partial products per fp32-accurate product" if split
                              else "f32-input MFMA, 64 FLOP/clk/SIMD"),
                "achieved_vs_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "mfma_issue_tflops_16bit": round(ach * nprod, 1) if split else None,
                "flops_per_launch": flops, "avg_launch_ms": round(avg * 1e3, 4), "launches_timed": len(dur), "traffic": None}
        eng.kernel_timers = None
        # per-stage roofline of the HBM-bound shell: algorithmic bytes / HIP-event duration vs 8 TB/s
        shell = {}
        for name, a, b, nbytes in list(eng.shell_timers or []) + aug_timers:
            t = shell.setdefault(name, [0, 0.0, 0])
            t[0] += 1; t[1] += a.elapsed_time(b) * 1e-3; t[2] += nbytes
        eng.shell_timers = None
        roof["shell"] = [{"kernel": k, "bound": "hbm", "achieved": round(v[2] / v[1] / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                          "frac": round(v[2] / v[1] / 8e12, 4), "avg_launch_ms": round(v[1] / v[0] * 1e3, 4),
                          "algorithmic_MB_per_launch": round(v[2] / v[0] / 1e6, 1), "launches_timed": v[0]} for k, v in shell.items()]
        if split and rank == 0:
            # what the matrix cores of THIS box sustain on random operands (clocks are power-limited and data-dependent):
            # register-only v_mfma_f32_32x32x16_bf16 loop, 8 waves/CU, measured right here (tools/micro/mfma_peak.hip)
            sus = measure_sustained_mfma()
            if sus:
                roof["mfma_sustained_measured"] = {"bf16_tflops": round(sus, 1), "split_equiv_tflops": round(sus / nprod, 1),
                                                   "frac_of_sustained": round(ach / (sus / nprod), 4),
                                                   "how": "tools/micro/mfma_peak.hip: register-only MFMA loop, random operands, 8 waves/CU x 4 blocks"}
        # the same kernel's average duration in the tracked rocprofv3 --kernel-trace --stats run of this command (profiles/CURRENT.json):
        # both fractions are printed so that the bench line and the profile can be compared directly
        rp = current_profile("rocprof_dominant")
        if rp and B == 32 and S == 768 and split and args.mode == "image" and json.load(open(rp)).get("arith", 3) == eng.arith and dom_is_bottleneck:
            rj = json.load(open(rp))
            roof["frac_rocprof"] = round(flops / (rj["avg_ms"] * 1e-3) / 1e12 / peak, 4)
            roof["rocprof"] = {"avg_launch_ms": rj["avg_ms"], "launches": rj["calls"], "source": rj["source"]}
        pmc = current_profile("pmc_dominant")
        if pmc and B == 32 and S == 768 and split and json.load(open(pmc)).get("arith", 3) == eng.arith and dom_is_bottleneck:     # counters were collected on this exact workload
            pj = json.load(open(pmc))
            # HBM bytes per launch of the dominant kernel from the PMC passes (read + write), then the break-down
            roof["traffic"] = round((pj["fetch_mb_per_launch"] + pj["write_mb_per_launch"]) * 1e6)
            roof["traffic_unit"] = "bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), from the tracked PMC pass in profiles/ -- not re-measured in this run"
            roof["traffic_detail"] = {"hbm_read_MB": pj["fetch_mb_per_launch"], "hbm_write_MB": pj["write_mb_per_launch"],
                                      "algorithmic_MB": pj["algorithmic_mb_per_launch"], "mfma_busy_pct_pmc": pj["mfma_busy_pct"],
                                      "source": pj["source"]}

    if roof is None and not args.no_kernel_timers:       # no bottleneck conv in this workload (detect only): whole-step fraction only
        roof = {"bound": "mfma", "kernel": None, "unit": "TFLOP/s", "peak": round(peak_split(eng) if eng.use_split else PEAK_F32_MFMA_TFLOPS, 1)}
    allgather_ms = host = None
    if dist_on:          # host side of a rank (8 Python ranks per node: resident set and time from process start to the end of the warm-up), max over ranks
        import resource
        rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6          # (Linux: kB)
        hv = torch.tensor([rss_gb, t_ready - T_PROCESS_START], device=coll_device(dev), dtype=torch.float64)
        torch.distributed.all_reduce(hv, op=torch.distributed.ReduceOp.MAX)
        host = {"rss_gb_max_over_ranks": round(float(hv[0]), 2), "startup_s_max_over_ranks": round(float(hv[1]), 1),
                "note": "per rank: peak resident set, seconds from process start to the first timed step (imports, random-init weights, packing, warm-up)"}
    if dist_on:          # the one collective of the path, timed on its own: RCCL all-gather of the [frames, 1 + nbits] logits
        n_tot = args.frames if stream else B * world
        loc = preds[f0:f1] if (stream and preds.shape[0] == n_tot) else preds[: (f1 - f0) if stream else B]
        loc = loc.contiguous()
        barrier()
        t1 = time.perf_counter()
        for _ in range(20):
            gather_frame_logits(loc, n_tot, align=(16 if stream else B))
        barrier()
        allgather_ms = (time.perf_counter() - t1) / 20 * 1e3
    if rank == 0:
        total_frames = (args.frames if stream else B * world) * args.steps
        fps = total_frames / elapsed
        # algorithmic work per frame: 28.28 GMAC embed (image) / 7.07 (video, step 4) + 6.16 GMAC detect
        emb_gmac = {"videoseal_1.0": 28.28, "pixelseal": 59.65}.get(args.card)        # BASELINE.md 2 (embedder GMAC per 256 x 256 pass)
        gmac = ((emb_gmac or 0.0) if not is_video else (emb_gmac or 0.0) / cfg.step_size) + 6.16
        if args.detect_only:
            gmac = 613.6 if args.card == "chunkyseal" else 6.16
        known_macs = emb_gmac is not None or (args.card == "chunkyseal" and args.detect_only)     # networks BASELINE.md 2 / SURVEY 8(d) count
        if not known_macs:
            gmac = 0.0
        if roof is not None and known_macs:
            roof["e2e_frac"] = round(fps * gmac * 2e9 / 1e12 / world / (peak_split(eng) if eng.use_split else PEAK_F32_MFMA_TFLOPS), 4)
            roof["e2e_note"] = "whole-step dense conv/GEMM FLOP/s (SURVEY 8(d) per-frame GMAC x frames/s) over the same MFMA ceiling"
        metric = f"frames/sec embed+extract {cfg.nbits}-bit @{S}x{S}"
        if args.detect_only:
            metric = f"frames/sec extract ({args.card}) @{S}x{S}"
        elif chain:
            metric = f"frames/sec embed+augment+extract {cfg.nbits}-bit @{S}x{S} (BASELINE configs[2])"
        line = {
            "metric": metric, "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "host_issue_ms_per_step": round(1e3 * host_issue / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if stream else "weak", "vs_baseline": None,
            "dtype": f"f32 ({arith_name(eng)}, fp32 accumulate)" if eng.use_split else "f32", "data": "synthetic",
            "config": {"workload": f"{args.card} {cfg.nbits}-bit, {B} frames {S}x{S} per GPU, {args.mode} mode "
                                   f"({'embedder on every frame' if not is_video else 'key frames every %d' % cfg.step_size}, "
                                   f"{'low-res' if args.lowres_attenuation else 'full-res'} JND), " + ("detect only" if args.detect_only else "embed + detect")
                                   + (", all-gather of bit logits" if dist_on else "")
                                   + (f"; streaming: {args.frames}-frame clip in 16-frame chunks" + (", one embed + detect call per chunk" if (args.group == 1 or not args.overlap) else f", key frames of {args.group or 8} chunks per U-Net pass, extractor on {args.det_batch or 128} frames per pass, watermark expanded and handed on chunk by chunk") + ", low-res JND" + (", uint8 RGB24 in/out" if args.u8 else "") + (", detect overlapped with the next embed on a second stream" if args.overlap else "") if stream else "")
                                   + (", chain JPEG(40) -> Crop(0.71) -> Resize(0.71) -> Brightness(0.5) -> Contrast(1.5) -> Saturation(1.5) -> Hue(0.1) between embed and detect (augmentation.Sequential: Crop + Resize + Brightness one kernel, Contrast + Saturation + Hue one pass)" if chain else "")
                                   + (", hipGraph replay" if args.graphs else ""),
                       "card": args.card, "weights": "random-init (seeded), no checkpoint offline", "batch_per_gpu": B,
                       "frame": [S, S], "mode": args.mode},
            "model_tflops_per_s": (round(fps * gmac * 2e9 / 1e12 / world, 2) if known_macs else None),
            "value_pipelined": (round(pipelined, 2) if pipelined else None),
            "value_pipelined_note": "the same steps with detect(batch i) on a second HIP stream under embed(batch i+1) (bench.py --pipeline); `value` is the sequential run",
            "roofline": roof,
            "arith_used": arith_used(eng),
        }
        if allgather_ms is not None:
            line["allgather_ms"] = round(allgather_ms, 4)
            line["n_ranks_seen"] = n_ranks_seen
            line["host"] = host
            line["collective"] = ("gloo over pinned host buffers (ranks may share a device: VS_BENCH_COLLECTIVE=gloo)"
                                  if torch.distributed.get_backend() == "gloo" else "RCCL all_gather_into_tensor")
            line["shards"] = ([list(shard_range(args.frames, r, world, 16)) for r in range(world)] if stream else [[r * B, (r + 1) * B] for r in range(world)])
        if args.dump_preds:          # the gathered logits of the last step, for a caller that recomputes them (tests/test_gpu_zdist.py)
            torch.save(preds.detach().cpu(), args.dump_preds)
        if not args.no_cpu_baseline and world == 1 and not args.detect_only:
            card_path = os.path.join(ROOT, "videoseal_amd", "cards", args.card + ".yaml")
            line["cpu_baseline"] = cpu_baseline(card_path, S, args.mode, cfg.step_size)
        else:
            line["cpu_baseline"] = None
        return line
    return None


def gen_step_leg(dev):
    """the generator-side training step (train.py:626-643 with the published recipe's terms: decoding + 0.1 x yuv perceptual, fixed weights)
    on VideoSeal 1.0, 16 frames of 256x256: forward + loss + backward on the HIP path (videoseal_amd.training.GeneratorStep)"""
    import videoseal_amd
    from videoseal_amd.training import GeneratorStep
    model = videoseal_amd.build("videoseal_1.0", seed=0).to(dev).train()
    B = 16
    frames = synthetic_batch(B, 256, dev, seed=7)
    masks = torch.ones(B, 1, 256, 256, device=dev)
    msgs = torch.randint(0, 2, (B, model.embedder.cfg.nbits), generator=torch.Generator().manual_seed(5))
    gs = GeneratorStep(model, percep_loss="yuv", percep_weight=0.1, decode_weight=1.0, balanced=False)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        gs.step(frames, masks, msgs)
    torch.cuda.synchronize()
    K = 5
    # two passes of K steps, the SECOND is reported: the first pass of this leg in a process that has already run the inference legs is
    # 50 % slower ON THE HOST SIDE (65 vs 43 ms per step, host issue time 62 ms; no device malloc / free happens in it -- the cause is not
    # isolated, the allocator re-cutting the inference-sized blocks it holds is the suspect), a second pass is at the rate of a fresh
    # process (tools/diag_train_leg.py, profiles/r03z_end_train_leg_warmup.log)
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(K):
            model.zero_grad(set_to_none=True)
            gs.step(frames, masks, msgs)
        t_issue = (time.perf_counter() - t0) / K      # how long the host needs to issue a step (the GPU runs behind it)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
    gflop = 3 * 2 * (28.28 + 6.16) * B               # forward + backward-data + backward-weights of the dense conv / GEMM work
    return {"value": round(B / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 2), "host_issue_ms": round(t_issue * 1e3, 2),
            "workload": "videoseal_1.0 generator step (forward + decoding / yuv loss + backward of embedder AND extractor), 16 x 256x256, fp32 gradients",
            "model_tflops_per_s": round(gflop / dt / 1e3, 1)}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-execute under torch.distributed.run, one rank per GPU of this node
    (rendezvous on 127.0.0.1: the container hostname may not resolve).  Too few GPUs: ONE JSON error line, exit code 2."""
    import socket
    import subprocess
    n = torch.cuda.device_count()
    if n < args.gpus:
        print(json.dumps({"error": f"--gpus {args.gpus} but this node shows {n} GPU(s)", "n_gpus_requested": args.gpus, "n_gpus_visible": n}), flush=True)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.cpu_baseline_only:
        import videoseal_amd           # (card -> step size; no device is touched)
        from videoseal_amd.layout import cfg_from_card, load_card
        card_path = os.path.join(ROOT, "videoseal_amd", "cards", args.card + ".yaml")
        print(json.dumps(cpu_baseline(card_path, args.size, args.mode, cfg_from_card(load_card(card_path)).step_size)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("VS_BENCH_FORCE_DIST") != "1":
        self_launch(args)
    line = run(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    default_run = (args.mode == "image" and args.card == "videoseal_1.0" and args.size == 768 and args.batch in (None, 32) and not args.capi
                   and not args.detect_only and not args.graphs and not args.pipeline and not args.no_extra)
    if default_run:
        # the other BASELINE configs as short legs of the same command, each with its own roofline fractions
        def leg(extra, steps=20, warmup=3):
            a = parse_args(["--gpus", str(args.gpus), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-extra"] + extra)
            # the previous leg's model and workspace are garbage by now, but torch's caching allocator keeps their blocks RESERVED in this process:
            # hand them back before the next leg builds its own (ChunkySeal's 1.8 B parameters + packed images are ~33 GB per rank; with the
            # one-device preflight's eight ranks sharing 288 GB the reserved leftovers of the earlier legs were the difference to an out-of-memory)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            try:
                r = run(a)
            except Exception as e:          # the headline line must survive a failing extra leg (one process: with several ranks a
                if world > 1:               # rank that carried on alone would leave the others in a collective -- let the launcher end the job)
                    raise
                return {"error": repr(e)[:300]}
            if r is None:
                return None
            roof = r.get("roofline") or {}
            return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "host_issue_ms_per_step": r.get("host_issue_ms_per_step"), "scaling": r["scaling"], "n_gpus": r["n_gpus"],
                    "allgather_ms": r.get("allgather_ms"), "host": r.get("host"), "arith_used": r.get("arith_used"),
                    "workload": r["config"]["workload"], "model_tflops_per_s": r["model_tflops_per_s"],
                    "roofline": {k: roof.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "e2e_frac") if k in roof},
                    "shell": roof.get("shell")}
        legs = {}
        if world == 1:
            legs["image 32x256x256 (north_star's 256x256 clips)"] = leg(["--size", "256"])
            legs["image 32x768 on the exact 3 x bf16 split (what the range guard falls back to: worst-case arithmetic)"] = leg(["--conv", "bf16x3"])
            legs["video_step4 (configs[1], video mode)"] = leg(["--mode", "video"])
            legs["chain (configs[2])"] = leg(["--mode", "chain"])
        legs["stream_1024 (configs[3], strong scaling over the ranks)"] = leg(["--mode", "stream", "--frames", "1024"], steps=4, warmup=1)
        # (ranks SHARING a device -- the one-device preflight of tests/test_gpu_zdist.py, VS_BENCH_COLLECTIVE=gloo -- take 4 frames per rank: eight
        # ranks x (21.6 GB of ChunkySeal weights + packed images + the 16-frame workspace) is 272 of the device's 288 GB, too close to an out-of-memory
        # for a plumbing test; one rank per GPU runs the stated 16)
        shared = os.environ.get("VS_BENCH_COLLECTIVE", "nccl") == "gloo" and world > max(1, torch.cuda.device_count())
        legs["chunkyseal_detect_16x1024 (configs[4])"] = leg(["--card", "chunkyseal", "--size", "1024", "--batch", "4" if shared else "16", "--detect-only"], steps=5, warmup=1)
        if world == 1:      # the other released cards on the configs[1] workload (no MAC count from the survey: frames/s only)
            legs["pixelseal image mode 32x768"] = leg(["--card", "pixelseal"], steps=10, warmup=2)
            legs["videoseal_0.0 (RMSNorm U-Net + ViT extractor) image mode 32x768"] = leg(["--card", "videoseal_0.0"], steps=10, warmup=2)
        if world == 1 and rank == 0:
            try:
                legs["train_step"] = gen_step_leg(torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
            except Exception as e:          # the bench line must survive a failing extra leg
                legs["train_step"] = {"error": repr(e)[:300]}
        if line is not None:
            line["configs"] = legs
    if line is not None:
        print(json.dumps(line), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
