"""Clip-level driver for the streaming configuration (inference_streaming.py:83-164): a long clip goes through the model as
16-frame `embed(..., is_video=True, lowres_attenuation=True)` + `detect` calls.  With 4 key frames / 16 frames per call most
kernels cannot fill 256 CUs on their own, so detect(chunk i) is issued on a second HIP stream while embed(chunk i+1) runs on the
first.  The calls, their arguments and their results are unchanged; only their placement on streams differs.
Safe because: embedder and extractor use disjoint named workspace buffers, K-split workspaces are per stream, every watermarked
chunk is a fresh tensor (recorded on the consuming stream), and the extractor's logits are cloned on the detect stream."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def embed_detect_chunks(model, frames: torch.Tensor, msgs: torch.Tensor, chunk: int = 16, lowres_attenuation: bool = True,
                        overlap: bool = True, sink: Optional[Callable[[int, torch.Tensor], None]] = None) -> torch.Tensor:
    """frames [F,3,H,W] fp32 or uint8 [F,H,W,3] on the device -> logits [F, 1+nbits].  `sink(first_frame, imgs_w_chunk)` receives
    every watermarked chunk (e.g. to hand it to an encoder); it is called on the embed stream's timeline."""
    u8 = frames.dtype == torch.uint8
    emb = (lambda x: model.embed_u8(x, msgs, lowres_attenuation=lowres_attenuation)) if u8 else \
          (lambda x: model.embed(x, msgs, is_video=True, lowres_attenuation=lowres_attenuation))
    det = (lambda w: model.detect_u8(w)) if u8 else (lambda w: model.detect(w, is_video=True))
    F_ = frames.shape[0]
    logits = []
    if not overlap:
        for a in range(0, F_, chunk):
            w = emb(frames[a:a + chunk])["imgs_w"]
            if sink:
                sink(a, w)
            logits.append(det(w)["preds"])
        return torch.cat(logits, 0)
    cur = torch.cuda.current_stream()
    s_emb, s_det = _streams(frames.device)
    s_emb.wait_stream(cur)
    s_det.wait_stream(cur)
    for a in range(0, F_, chunk):
        with torch.cuda.stream(s_emb):
            w = emb(frames[a:a + chunk])["imgs_w"]
            if sink:
                sink(a, w)
            ev = torch.cuda.Event()
            ev.record(s_emb)
        with torch.cuda.stream(s_det):
            s_det.wait_event(ev)
            w.record_stream(s_det)
            logits.append(det(w)["preds"])
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
