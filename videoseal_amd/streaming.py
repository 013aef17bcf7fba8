"""Clip-level driver for the streaming configuration (inference_streaming.py:83-164): a long clip goes through the model as
16-frame `embed(..., is_video=True, lowres_attenuation=True)` + `detect` calls.  With 4 key frames / 16 frames per call the matrix
kernels cannot fill 256 CUs (the U-Net bottleneck conv runs at 0.19 of its ceiling instead of 0.53), so when the clip is resident:

  * the key frames of `group` consecutive chunks go through the U-Net as ONE batch (`Videoseal.embed_group`; default: enough chunks for
    32 key frames), the watermark is expanded chunk by chunk as the per-chunk calls would do it, and `sink` still sees chunk after chunk;
  * the extractor runs on up to DET_BATCH (128) watermarked frames at a time, whatever the caller's chunk;
  * detect(group i) is issued on a second HIP stream while embed(group i+1) runs on the first, and (round 6) starts on the first 32
    watermarked frames of a group while the tail of the remaining frames is still being issued.

`group=1` is exactly the sequence of per-chunk calls (bit-identical; the round-3 behaviour); larger groups differ from it only by the
summation order of the dense layers, because whether K is split is a function of the batch shape (engine._split_k_rule*).
Stream safety: embedder and extractor use disjoint named workspace buffers, K-split workspaces are per stream, every watermarked
group is a fresh tensor (recorded on the consuming stream), and the extractor's logits are cloned on the detect stream."""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch

KEY_BATCH = 32      # key frames per U-Net pass at which conv3x3_pl_kernel has one 256-pixel tile x 192 channels per CU (B * 4 * 2 = 256)
DET_BATCH = 128     # frames per extractor pass.  Round 6: 32 -> 128 (= one default group of 8 x 16 frames): the extractor's stage-2 / 3 GEMMs are one-round
                    # launches at 32 frames (256 tiles on 256 CUs, every launch pays its fill, epilogue and store drain once per 32 frames); same box,
                    # bench.py --detect-only: 2.96 ms per 32 frames at batch 32, 2.61 at batch 64; stream leg 1024 frames 7586 / 8189 / 8332 frames/s and a
                    # 128-frame shard 7243 / 7857 / 8248 frames/s at 32 / 64 / 128 frames per pass (profiles/r06f_det_batch_sweep.json)


GROUP_BYTES = 2 << 30   # source + watermarked frames of one group (a 768 x 768 fp32 group of 8 x 16 frames is 1.8 GB; 4K fp32 chunks go one by one)


def default_group(chunk: int, step: int, frame_bytes: int = 0) -> int:
    """chunks per U-Net pass: enough for KEY_BATCH key frames, capped so that the group's source + output frames stay under GROUP_BYTES
    (`frame_bytes` = bytes of one frame as passed in); 1 (= the per-chunk calls) when the chunks' key frames are not the group's
    every-step-th frames"""
    if chunk % step:
        return 1
    g = max(1, (KEY_BATCH * step + chunk - 1) // chunk)
    if frame_bytes > 0:
        g = max(1, min(g, GROUP_BYTES // (2 * chunk * frame_bytes)))
    return g


def embed_detect_chunks(model, frames: torch.Tensor, msgs: torch.Tensor, chunk: int = 16, lowres_attenuation: bool = True,
                        overlap: bool = True, sink: Optional[Callable[[int, torch.Tensor], None]] = None,
                        group: Optional[int] = None, det_batch: Optional[int] = None) -> torch.Tensor:
    """frames [F,3,H,W] fp32 or uint8 [F,H,W,3] on the device -> logits [F, 1+nbits].  `sink(first_frame, imgs_w_chunk)` receives
    every watermarked chunk in clip order (e.g. to hand it to an encoder); it is called on the embed stream's timeline.
    group: chunks per U-Net pass (None = default_group; 1 = the literal per-chunk calls); det_batch: frames per extractor pass (None = DET_BATCH)."""
    u8 = frames.dtype == torch.uint8
    step = int(model.step_size)
    if group is None:
        group = default_group(chunk, step, frames[0].numel() * frames.element_size() if frames.shape[0] else 0)
        if chunk > int(model.chunk_size) * step:          # embed() would split such a chunk internally: the literal per-chunk calls
            group = 1
    if group > 1 and chunk % step:
        raise ValueError(f"group > 1 needs chunk ({chunk}) to be a multiple of step_size ({step})")
    F_ = frames.shape[0]
    span = chunk * group

    def emb(x):
        if group == 1:       # the caller's own calls
            return (model.embed_u8(x, msgs, lowres_attenuation=lowres_attenuation) if u8 else
                    model.embed(x, msgs, is_video=True, lowres_attenuation=lowres_attenuation))["imgs_w"]
        return model.embed_group(x, msgs, chunk, lowres_attenuation=lowres_attenuation)

    def det(w):
        # detect_u8 / detect(is_video=True) walk `w` in model.chunk_size frames; raise it to DET_BATCH for this call so that the extractor
        # sees full batches (frames are independent: the logits of a frame do not depend on the batch it is in, up to the K-split rule)
        old = model.chunk_size
        model.chunk_size = max(int(old), int(det_batch or DET_BATCH)) if group > 1 else old
        try:
            return (model.detect_u8(w) if u8 else model.detect(w, is_video=True))["preds"]
        finally:
            model.chunk_size = old

    def feed_sink(a, w):
        if sink:
            for c in range(0, w.shape[0], chunk):
                sink(a + c, w[c:c + chunk])

    logits = []
    if not overlap:
        for a in range(0, F_, span):
            w = emb(frames[a:a + span])
            feed_sink(a, w)
            logits.append(det(w))
        return torch.cat(logits, 0)
    cur = torch.cuda.current_stream()
    s_emb, s_det = _streams(frames.device)
    s_emb.wait_stream(cur)
    s_det.wait_stream(cur)
    db = int(det_batch or DET_BATCH)
    # round 6: inside a group the extractor starts on the first `db` watermarked frames while the tail of the rest is still being issued
    # (`embed_group(on_tail=...)`): a rank that holds ONE group (128 of the 1024 frames on 8 GPUs) has no "next embed" to hide detect under, so
    # the only overlap it can have is inside the group.  Same U-Net batch, same extractor batches -> the logits are bit-identical to the
    # whole-group hand-over (`fine=False`); VIDEOSEAL_STREAM_FINE=0 restores that
    fine = (group > 1 and db % chunk == 0 and db % step == 0 and not getattr(model, "use_graphs", False)
            and os.environ.get("VIDEOSEAL_STREAM_FINE", "1") != "0")
    for a in range(0, F_, span):
        if fine:
            def piece(first, wp):             # on the embed stream's timeline, right behind the tail launches of wp
                ev = torch.cuda.Event()
                ev.record(s_emb)
                with torch.cuda.stream(s_det):
                    s_det.wait_event(ev)
                    wp.record_stream(s_det)
                    logits.append(det(wp))
            with torch.cuda.stream(s_emb):
                w = model.embed_group(frames[a:a + span], msgs, chunk, lowres_attenuation=lowres_attenuation, on_tail=piece, tail_batch=db)
                feed_sink(a, w)
            continue
        with torch.cuda.stream(s_emb):
            w = emb(frames[a:a + span])
            feed_sink(a, w)
            ev = torch.cuda.Event()
            ev.record(s_emb)
        with torch.cuda.stream(s_det):
            s_det.wait_event(ev)
            w.record_stream(s_det)
            logits.append(det(w))
    cur.wait_stream(s_emb)
    cur.wait_stream(s_det)
    out = torch.cat(logits, 0)
    for t in logits:
        t.record_stream(cur)
    return out


_STREAMS = {}

def _streams(device) -> Tuple[torch.cuda.Stream, torch.cuda.Stream]:
    key = str(device)
    if key not in _STREAMS:
        _STREAMS[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _STREAMS[key]
