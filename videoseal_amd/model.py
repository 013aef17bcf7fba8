"""The ``Videoseal`` nn.Module surface on top of the HIP engine.

Drop-in boundary: same constructor products, attributes, method signatures, state_dict
keys and error behaviour as the reference's ``videoseal.models.Videoseal`` (models/videoseal.py:15-428,
models/wam.py:18-234) for the inference path ``embed / detect / extract_message`` plus the forward-only
``forward``; all arithmetic runs in hand-written gfx950 kernels (engine.py -> native.py -> csrc/).

Deliberate differences, each loud rather than silent:
  * no CPU / ATen execution path: the model must live on a ROCm device (``.to('cuda')``);
  * ``forward`` is differentiable (videoseal_amd/autograd.py): with autograd enabled and trainable parameters it returns tensors
    whose graph nodes run the HIP backward kernels, so train.py's `loss.backward()` / `torch.autograd.grad(loss, last_layer)` / DDP hooks
    work unchanged; under ``torch.no_grad()`` (or with everything frozen) it is the values-only launch sequence;
  * frames handed over on the CPU are copied to the model's device, processed there end to end and the
    results are copied back to ``imgs.device`` (the reference keeps the full-resolution shell on the CPU); for a CPU caller the
    no-grad entry points return tensors from torch's caching pinned allocator (``_result_buffer``).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import native as N
from .engine import Act, HipEngine
from .layout import ModelCfg, build_tree, detector_entries, embedder_entries

_DEFAULT_INTERP = {"mode": "bilinear", "align_corners": False, "antialias": True}


_PINNED_RESULTS = os.environ.get("VIDEOSEAL_PINNED_RESULTS", "1") != "0"


def _result_buffer(shape, dtype, device) -> torch.Tensor:
    """Where a result goes when the caller's frames are not on the model's device.  For a CPU caller (the calling convention of
    videoseal.py:286-297 / inference_av.py: the clip stays on the CPU) the tensor comes from torch's caching PINNED allocator: the copy
    back runs at link speed instead of through a freshly mapped pageable tensor (32 x 768^2 fp32 frames: 4 ms instead of ~40 ms,
    tools/bench_pcie.py).  It is an ordinary CPU tensor for the caller; VIDEOSEAL_PINNED_RESULTS=0 returns pageable memory."""
    device = torch.device(device)
    if device.type == "cpu" and _PINNED_RESULTS:
        try:
            return torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
        except RuntimeError:          # the host refuses to lock that much memory: pageable result, same values
            pass
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def _to_caller(t, device):
    """device tensor -> the caller's device (no-grad result paths)"""
    if not torch.is_tensor(t) or t.device == torch.device(device):
        return t
    if torch.device(device).type != "cpu" or not t.is_cuda:
        return t.to(device)
    out = _result_buffer(t.shape, t.dtype, device)
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return out


def _antialias_flag(interpolation: Optional[dict]) -> bool:
    it = dict(_DEFAULT_INTERP if interpolation is None else interpolation)
    if it.get("mode", "bilinear") != "bilinear" or it.get("align_corners", False):
        raise NotImplementedError(f"interpolation {it}: the HIP path implements mode='bilinear', align_corners=False "
                                  f"(antialias True/False)")
    return bool(it.get("antialias", False))


class _EngineOwner:
    """Lazily (re)builds the packed-weight engine when parameters, device or mode change."""

    def _engine_for(self, root: "Wam") -> HipEngine:
        return root._engine()


class Embedder(nn.Module):
    """``model.embedder(imgs01, msgs) -> delta`` (models/embedder.py:130-165).  Holds ``unet`` / ``msg_processor``
    parameter trees; ``yuv`` tells Wam to feed the Y channel (embedder.py:281)."""

    def __init__(self, cfg: ModelCfg, seed: int = 0):
        super().__init__()
        tree = build_tree(embedder_entries(cfg), seed)
        self.unet = tree.unet
        self.msg_processor = tree.msg_processor
        self.cfg = cfg
        self.yuv = cfg.yuv
        self._root = None      # set by Wam (plain attribute holder, avoids a module cycle)

    def get_random_msg(self, bsz: int = 1, nb_repetitions: int = 1) -> torch.Tensor:
        k = self.cfg.nbits                                   # msg_processor.py:45-59
        if nb_repetitions != 1:
            assert k % nb_repetitions == 0, f"nbits must be divisible by nb_repetitions, got {k} and {nb_repetitions}"
            aux = torch.randint(0, 2, (bsz, k // nb_repetitions))
            return aux.unsqueeze(1).repeat(1, nb_repetitions, 1).view(bsz, k)
        return torch.randint(0, 2, (bsz, k))

    def get_last_layer(self) -> torch.Tensor:
        return self.unet.outc.weight                         # embedder.py:147-149

    def forward(self, imgs: torch.Tensor, msgs: torch.Tensor) -> torch.Tensor:
        root = self._root[0]
        eng = root._engine()
        with torch.cuda.device(eng.dev):
            x = N.f32c(imgs.to(eng.dev))
            B, Cc, H, W = x.shape
            if Cc != self.cfg.in_ch:
                raise ValueError(f"embedder expects {self.cfg.in_ch} input channel(s), got {Cc}")
            ident = (1.0, 0.0, 0.0)
            import ctypes as C
            key = eng.new_act("emb.in", B, H, W, Cc, 4)
            ymat = (C.c_float * 3)(*ident) if Cc == 1 else None
            N.check(eng.lib.vs_resize_pre(N.ptr(x), B, Cc, H, W, H, W, 0, None, 1.0, 0.0, N.ptr(key.t), 1, ymat, N.stream()), "vs_resize_pre")
            delta = eng.embedder_forward(key, _msgs_i32(msgs, eng.dev), bn_train=self.training)
            return delta.clone().to(imgs.device)


class Extractor(nn.Module):
    """``model.detector(imgs01) -> logits [b, 1+nbits]`` (models/extractor.py:140-167)."""

    def __init__(self, cfg: ModelCfg, seed: int = 0):
        super().__init__()
        tree = build_tree(detector_entries(cfg), seed + 1)
        if cfg.extractor == "sam":            # SegmentationExtractor (extractor.py:40-75): image_encoder + pixel_decoder
            self.image_encoder = tree.image_encoder
        else:
            self.convnext = tree.convnext
        self.pixel_decoder = tree.pixel_decoder
        self.cfg = cfg
        self._root = None

    def forward(self, imgs: torch.Tensor) -> torch.Tensor:
        root = self._root[0]
        eng = root._engine()
        with torch.cuda.device(eng.dev):
            x = N.f32c(imgs.to(eng.dev))
            rgb, _ = eng.resize_pre(x, (x.shape[-2], x.shape[-1]), False, want_rgb=True, mul=2.0, add=-1.0, tag="det.in")
            return eng.extractor_forward(rgb).clone().to(imgs.device)


class Blender(nn.Module):
    """models/blender.py:12-68: only the additive mode is used by the released cards."""
    AVAILABLE_BLENDING_METHODS = ["additive"]

    def __init__(self, scaling_i: float, scaling_w: float, method: str = "additive"):
        super().__init__()
        if method != "additive":
            raise NotImplementedError(f"blending method '{method}': only 'additive' is implemented in the HIP path")
        self.method, self.scaling_i, self.scaling_w = method, scaling_i, scaling_w


class RGB2YUV(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("M", torch.tensor([[0.299, 0.587, 0.114], [-0.14713, -0.28886, 0.436], [0.615, -0.51499, -0.10001]],
                                               dtype=torch.float32))


class _Kernel2d(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.weight = nn.Parameter(k, requires_grad=False)


class JND(nn.Module):
    """modules/jnd.py:11-114 parameter holder (frozen 3x3 Sobel / 5x5 luminance taps live in the state_dict);
    ``heatmaps`` runs the HIP kernel."""

    def __init__(self, in_channels: int = 1, out_channels: int = 3):
        super().__init__()
        if (in_channels, out_channels) != (1, 1):
            raise NotImplementedError("JND: only in_channels=1, out_channels=1 (jnd_1_1, used by every released card)")
        self.in_channels, self.out_channels = in_channels, out_channels
        kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]])
        ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]])
        kl = torch.tensor([[1., 1., 1., 1., 1.], [1., 2., 2., 2., 1.], [1., 2., 0., 2., 1.], [1., 2., 2., 2., 1.], [1., 1., 1., 1., 1.]])
        self.conv_x = _Kernel2d(kx[None, None].repeat(in_channels, 1, 1, 1))
        self.conv_y = _Kernel2d(ky[None, None].repeat(in_channels, 1, 1, 1))
        self.conv_lum = _Kernel2d(kl[None, None].repeat(in_channels, 1, 1, 1))
        self._root = None

    def heatmaps(self, imgs: torch.Tensor, clc: float = 0.3) -> torch.Tensor:
        if clc != 0.3:
            raise NotImplementedError("clc != 0.3")
        eng = self._root[0]._engine()
        with torch.cuda.device(eng.dev):
            return eng.jnd_full(N.f32c(imgs.to(eng.dev))).to(imgs.device)


def get_dummy_augmenter() -> nn.Module:
    """augmentation/augmenter.py:48-57 ``get_dummy_augmenter()``: an Augmenter whose only op is Identity.  The reference builds it
    with masks={'kind': None}, i.e. its OpenCV-drawn MixedMaskEmbedder; here the mask embedder is the full mask of the training
    config (configs/all_augs.yaml:2-3 `kind: none`) -- assign ``augmenter.mask_embedder`` to plug any other callable."""
    from .augmentation import Augmenter
    return Augmenter(augs={"identity": 1}, augs_params={}, masks={"kind": "none"})


def _msgs_i32(msgs: torch.Tensor, dev) -> torch.Tensor:
    m = msgs.to(dev)
    if m.is_floating_point():
        m = m.long()            # msg_processor.py:92 `(indices + msg).long()` truncation
    return m.to(torch.int32).contiguous()


_EXTRACTOR_FIELDS = ("depths", "dims", "stem_stride", "extractor", "vit_dim", "vit_depth", "vit_heads", "vit_patch", "vit_window", "vit_global",
                     "vit_out", "vit_mlp_ratio", "vit_rel_pos")


def merge_cfg(embedder: "Embedder", detector: "Extractor", **own) -> ModelCfg:
    """One ModelCfg for the engine out of the halves `build_embedder` / `build_extractor` produced (videoseal_amd/builders.py, train.py:262-281)
    + the numbers the Wam / Videoseal constructor was given.  Models built from a card already share one cfg."""
    import dataclasses
    e, d = embedder.cfg, detector.cfg
    if e is d:
        return e
    if e.nbits != d.nbits:
        raise ValueError(f"embedder carries {e.nbits} bits, extractor {d.nbits}")
    return dataclasses.replace(e, **{f: getattr(d, f) for f in _EXTRACTOR_FIELDS}, **own)


class Wam(nn.Module):
    """Image path (models/wam.py:18-234)."""

    def __init__(self, embedder: Embedder, detector: Extractor, augmenter: nn.Module, attenuation: Optional[JND] = None,
                 scaling_w: float = 1.0, scaling_i: float = 1.0, clamp: bool = True, img_size: int = 256,
                 blending_method: str = "additive") -> None:
        super().__init__()
        if embedder.cfg is not detector.cfg:      # halves from build_embedder / build_extractor (train.py:262-305)
            jnd = (attenuation.in_channels, attenuation.out_channels) if attenuation is not None else (0, 0)
            embedder.cfg = detector.cfg = merge_cfg(embedder, detector, img_size=int(img_size), scaling_w=float(scaling_w), scaling_i=float(scaling_i),
                                                    blending_method=str(blending_method), jnd_in=jnd[0], jnd_out=jnd[1])
        self.embedder, self.detector, self.augmenter = embedder, detector, augmenter
        self.img_size = img_size
        self.rgb2yuv = RGB2YUV()
        self.blender = Blender(scaling_i, scaling_w, blending_method)
        self.attenuation = attenuation
        self.clamp = clamp
        self._eng: Optional[HipEngine] = None
        self._wkeys: Dict[str, tuple] = {}
        self._wgroups: Optional[Dict[str, list]] = None
        self._msg_cache: Optional[tuple] = None
        self._bn_sync = None          # set by videoseal_amd.dist.convert_sync_batchnorm: the BatchNorm exchange of distributed training
        self._emb_bwd = self._det_bwd = None      # training.EmbedderBackward / DetectorStep of the differentiable forward (built on first use)
        self._train_gen: Dict[str, int] = {}      # generation of the operands the HIP backward reads from the engine's workspace
        self._warned_no_backward = False
        # hipGraph replay of the per-chunk launch sequences (fixed chunk shapes, e.g. streaming callers): the ~300 kernel
        # launches of an embed / detect chunk are captured once per (shape, flags) and replayed with one launch
        self.use_graphs = os.environ.get("VIDEOSEAL_GRAPHS", "0") == "1"
        self._graphs: Dict[tuple, dict] = {}
        holder = [self]
        embedder._root = detector._root = holder
        if attenuation is not None:
            attenuation._root = holder

    def __setattr__(self, name, value):
        if name == "attenuation" and isinstance(value, JND):       # evals swap the attenuation module (evals/full.py:317-336)
            value._root = [self]
        if name in ("attenuation", "embedder", "detector", "rgb2yuv") and "_wgroups" in self.__dict__:
            self.__dict__["_wgroups"] = None
        super().__setattr__(name, value)

    # ---- plumbing
    @property
    def device(self):
        return next(self.parameters()).device

    def get_random_msg(self, bsz: int = 1, nb_repetitions=1) -> torch.Tensor:
        return self.embedder.get_random_msg(bsz, nb_repetitions)

    def _apply(self, fn, *a, **k):                 # .to() / .cuda() / .float(): tensors are replaced -> regroup
        self._wgroups = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._wgroups = None
        return super().load_state_dict(*a, **k)

    def _weight_groups(self) -> Dict[str, list]:
        """the tensors each packed weight group is built from (cached: walking the module tree costs 0.6 ms per call)"""
        if self._wgroups is None:
            emb_p = list(self.embedder.parameters())
            emb_b = list(self.embedder.buffers())
            det = list(self.detector.parameters()) + list(self.detector.buffers())
            misc = list(self.rgb2yuv.buffers()) + (list(self.attenuation.parameters()) if self.attenuation is not None else [])
            self._wgroups = {"Et": emb_p, "E": emb_p + emb_b, "X": det, "misc": misc}
        return self._wgroups

    def _engine(self) -> HipEngine:
        dev = self.device
        if dev.type != "cuda":
            raise N.NativeError("Videoseal (MI355X build) has no CPU execution path: move the model to a ROCm device "
                                "with .to('cuda') before embed()/detect().")
        if self._eng is None or self._eng.dev != dev:
            self._eng = HipEngine(self.embedder.cfg, self.state_dict, dev)
            self._wkeys = {}
            self._graphs.clear()
            self._emb_bwd = self._det_bwd = None
            self._train_gen = {}
        # packed weights go stale when any source tensor is replaced (data_ptr) or written in place (_version): every tensor of
        # the group is part of the key
        stale = []
        for name, ts in self._weight_groups().items():
            key = tuple((t.data_ptr(), t._version) for t in ts)
            if self._wkeys.get(name) != key:
                if name in self._wkeys:
                    stale.append(name)
                self._wkeys[name] = key
        if stale:
            self._eng.invalidate(*stale)
            self._graphs.clear()
        if self._bn_sync is None and self.embedder.training and os.environ.get("VIDEOSEAL_SYNC_BN", "auto") == "auto":
            # train.py:438-440 converts every BatchNorm to SyncBatchNorm when it runs distributed; nn.SyncBatchNorm.convert_sync_batchnorm finds
            # no nn.BatchNorm2d children here (the U-Net is one HIP launch sequence), so the same switch is made when a process group with
            # more than one rank exists at the first train-mode forward (VIDEOSEAL_SYNC_BN=0 keeps per-rank statistics)
            import torch.distributed as td
            if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
                from .dist import bn_all_reduce
                self._bn_sync = bn_all_reduce()
        self._eng.bn_sync = self._bn_sync
        self._eng.poll_nonfinite()
        return self._eng

    def repack(self) -> None:
        """Force re-packing of the weights (after edits that bypass tensor versioning, e.g. writes through raw pointers)."""
        self._wgroups = None
        if self._eng is not None:
            self._eng.invalidate()
        self._graphs.clear()

    def _msgs_dev(self, msgs: torch.Tensor, dev) -> torch.Tensor:
        """int32 device copy of the message; a CPU message that has not changed since the last call is not copied again
        (streaming callers pass the same [1, k] tensor for every chunk: one pageable H2D sync per call otherwise)."""
        if msgs.device == dev:
            return _msgs_i32(msgs, dev)
        c = self._msg_cache
        if c is not None and c[0] is msgs and c[1] == msgs._version and c[2].device == dev:
            return c[2]
        mi = _msgs_i32(msgs, dev)
        self._msg_cache = (msgs, msgs._version, mi)
        return mi

    def _graphed(self, key: tuple, ins: Dict[str, torch.Tensor], run, rekey=None):
        """Replay `run(static_inputs) -> dict of output tensors` from a hipGraph captured once per key.
        The first call runs eagerly twice (workspace allocation + tile autotune), then captures.  `rekey()` recomputes the key after the
        eager passes: the first of them verifies new weights and may switch the network (or single layers) to 3 x bf16; the graph records
        the kernels of the arithmetic in force at capture, so it is filed under the key a later call will compute (a stale key would
        capture a second graph on the next call)."""
        ent = self._graphs.get(key)
        if ent is None:
            static_in = {k: v.clone() for k, v in ins.items()}
            run(static_in)
            run(static_in)
            torch.cuda.synchronize()
            if rekey is not None:
                key = rekey()
                ent = self._graphs.get(key)
        if ent is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = run(static_in)
            ent = {"g": g, "in": static_in, "out": outs}
            self._graphs[key] = ent
        for k, v in ins.items():
            ent["in"][k].copy_(v)
        ent["g"].replay()
        return ent["out"]

    def _detect_frames(self, eng: HipEngine, fr: torch.Tensor, S, antialias: bool) -> torch.Tensor:
        def run(si):
            rgb, _ = eng.resize_pre(si["fr"], S, antialias, want_rgb=True, mul=2.0, add=-1.0, tag="det.in")
            return {"preds": eng.extractor_forward(rgb)}
        if self.use_graphs and not torch.cuda.is_current_stream_capturing():
            def key():           # the arithmetic is part of the key: network-wide split AND the layers the calibration pinned to the exact split
                return ("det", fr.dtype, tuple(fr.shape), tuple(S), antialias, id(eng), eng.arith_net["X"], tuple(sorted(eng.layer_arith.items())))
            out = self._graphed(key(), {"fr": fr}, run, rekey=key)["preds"].clone()
            eng.note_guard()           # the captured vs_check_finite has run: deliver its flag like an eager steady-state pass does
            return out
        return run({"fr": fr})["preds"].clone()

    # ---- core of embed: one chunk of frames on the device
    def _embed_frames(self, eng: HipEngine, fr: torch.Tensor, msgs_i32: torch.Tensor, out: torch.Tensor, *, step: int,
                      video_mode: int, antialias: bool, lowres: bool, preds_w: Optional[torch.Tensor] = None,
                      fwd_order: bool = False, tail_span: int = 0) -> None:
        bn_train = self.embedder.training
        if self.use_graphs and not bn_train and not torch.cuda.is_current_stream_capturing():
            key = ("emb", fr.dtype, tuple(fr.shape), tuple(msgs_i32.shape), step, video_mode, antialias, lowres, preds_w is not None, self.img_size,
                   self.clamp, float(self.blender.scaling_i), float(self.blender.scaling_w), self.attenuation is not None, fwd_order, tail_span, id(eng), eng.arith_net["E"])
            ent = self._graphs.get(key)
            if ent is None:
                sin = {"fr": fr.clone(), "msgs": msgs_i32.clone()}
                sout = torch.empty_like(sin["fr"])
                spw = torch.empty_like(preds_w) if preds_w is not None else None
                for _ in range(2):
                    self._embed_frames_eager(eng, sin["fr"], sin["msgs"], sout, step=step, video_mode=video_mode, antialias=antialias,
                                             lowres=lowres, preds_w=spw, fwd_order=fwd_order, tail_span=tail_span)
                torch.cuda.synchronize()
                if eng.arith_net["E"] != key[-1]:      # the verification pass switched the embedder to 3 x bf16: capture under the new key
                    return self._embed_frames(eng, fr, msgs_i32, out, step=step, video_mode=video_mode, antialias=antialias, lowres=lowres,
                                              preds_w=preds_w, fwd_order=fwd_order, tail_span=tail_span)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._embed_frames_eager(eng, sin["fr"], sin["msgs"], sout, step=step, video_mode=video_mode, antialias=antialias,
                                             lowres=lowres, preds_w=spw, fwd_order=fwd_order, tail_span=tail_span)
                ent = {"g": g, "in": sin, "out": sout, "pw": spw}
                self._graphs[key] = ent
            ent["in"]["fr"].copy_(fr)
            ent["in"]["msgs"].copy_(msgs_i32)
            ent["g"].replay()
            eng.note_guard()
            out.copy_(ent["out"])
            if preds_w is not None:
                preds_w.copy_(ent["pw"])
            return
        self._embed_frames_eager(eng, fr, msgs_i32, out, step=step, video_mode=video_mode, antialias=antialias, lowres=lowres,
                                 preds_w=preds_w, fwd_order=fwd_order, tail_span=tail_span)

    def _embed_frames_eager(self, eng: HipEngine, fr: torch.Tensor, msgs_i32: torch.Tensor, out: torch.Tensor, *, step: int,
                            video_mode: int, antialias: bool, lowres: bool, preds_w: Optional[torch.Tensor] = None,
                            fwd_order: bool = False, tail_span: int = 0, on_tail=None, tail_batch: int = 0) -> None:
        """tail_span > 0 (a multiple of `step`): `fr` is a GROUP of consecutive caller chunks of tail_span frames each.  The key frames of the
        whole group go through the U-Net as one batch (the matrix kernels want >= 32 key frames to fill 256 CUs), the watermark is then
        expanded chunk by chunk exactly as the per-chunk calls would do it (videoseal.py:303-344: the last key frame of a chunk has no
        successor in 'interpolate' mode), so the group equals the sequence of per-chunk calls up to the summation order of the dense layers."""
        S = (self.img_size, self.img_size)
        att = self.attenuation is not None
        rgb, key = eng.resize_pre(fr, S, antialias, want_rgb=(att and lowres), want_key=True, key_step=step)
        delta = eng.embedder_forward(key, msgs_i32, bn_train=self.embedder.training)
        hmap = eng.jnd_lowres(rgb) if (att and lowres) else None
        kw = dict(step=step, video_mode=video_mode, attenuate=(2 if (att and fwd_order and not lowres) else int(att)), clamp=self.clamp,
                  antialias=antialias, scaling_i=self.blender.scaling_i, scaling_w=self.blender.scaling_w)
        F_ = fr.shape[0]
        # on_tail(a, b) (streaming.py): called on this stream's timeline every time another `tail_batch` watermarked frames [a, b) have been
        # issued, so that the consumer (the extractor on a second stream) starts on the first frames of a group while the rest is still being
        # blended.  The tail is a per-frame operation given (delta, hmap): cutting it at chunk boundaries does not change a value
        cut = tail_batch if (on_tail is not None and tail_batch > 0) else 0
        if tail_span > 0 and tail_span % step:
            raise ValueError("tail_span must be a multiple of the key-frame step")
        if cut and cut % max(tail_span, step):
            raise ValueError(f"tail_batch ({tail_batch}) must be a multiple of the chunk ({tail_span}) and of the key-frame step ({step})")
        per_chunk = not (tail_span <= 0 or tail_span >= F_ or video_mode == N.VIDEO_MODES["repeat"])
        # 'repeat' looks at one key frame per output frame: one launch over the group is the per-chunk launches
        span = tail_span if per_chunk else (cut if cut else F_)
        hw = S[0] * S[1]
        done = 0
        for a in range(0, F_, span):
            b = min(F_, a + span)
            if a == 0 and b == F_:
                eng.embed_tail(fr, out, delta, hmap_low=hmap, preds_w=preds_w, **kw)
            else:
                eng.embed_tail(fr[a:b], out[a:b], delta[a // step:(b + step - 1) // step], hmap_low=(hmap[a * hw:b * hw] if hmap is not None else None),
                               preds_w=(preds_w[a:b] if preds_w is not None else None), **kw)
            if cut and (b - done >= cut or b == F_):
                on_tail(done, b)
                done = b
        if on_tail is not None and not cut:
            on_tail(0, F_)

    def _run_chunks(self, eng: HipEngine, imgs: torch.Tensor, span: int, fn, *, want_out: bool = True, extra=None):
        """Drive `fn(chunk_on_device, out_chunk_on_device, a, b)` over [a, b) frame ranges of `span` frames.  Frames that are not
        on the model's device move one chunk at a time (videoseal.py:286-297 keeps the full clip where the caller put it);
        the result comes back on imgs.device."""
        on_dev = imgs.device == eng.dev
        if on_dev:
            src = imgs if imgs.dtype == torch.uint8 else N.f32c(imgs)
            out = torch.empty_like(src) if want_out else None
        else:
            src = imgs
            out = _result_buffer(imgs.shape, imgs.dtype if imgs.dtype == torch.uint8 else torch.float32, imgs.device) if want_out else None
        for attempt in range(2):
            for a in range(0, imgs.shape[0], span):
                b = min(imgs.shape[0], a + span)
                if on_dev:
                    fn(src[a:b], out[a:b] if want_out else None, a, b)
                else:
                    ch = src[a:b].to(eng.dev)
                    ch = ch.contiguous() if ch.dtype == torch.uint8 else N.f32c(ch)
                    oc = torch.empty_like(ch) if want_out else None
                    fn(ch, oc, a, b)
                    if want_out:
                        out[a:b].copy_(oc, non_blocking=True)
            if on_dev:
                break
            # the copies into (pinned) host memory are asynchronous: the results are valid from here on.  This entry point synchronises
            # anyway, so the range guard of every pass of the call is read here: a data-dependent overflow repeats the call on the exact
            # split instead of handing non-finite frames to the caller
            torch.cuda.current_stream(eng.dev).synchronize()
            if attempt or not eng.guard_tripped_after_sync():
                break
            if extra is not None:
                extra()               # (the caller's per-call accumulators, e.g. the list of logits, start over)
        return out

    @torch.no_grad()
    def embed(self, imgs: torch.Tensor, msgs: torch.Tensor = None, interpolation: dict = None,
              lowres_attenuation: bool = False) -> dict:
        """wam.py:134-204."""
        if msgs is None:
            msgs = self.get_random_msg(imgs.shape[0])
        eng = self._engine()
        aa = _antialias_flag(interpolation)
        B = imgs.shape[0]
        cd = self.embedder.cfg.out_ch
        if B == 0:
            return {"msgs": msgs, "preds_w": imgs.new_zeros((0, cd) + tuple(imgs.shape[-2:])), "imgs_w": imgs.clone()}
        if msgs.shape[0] != B:
            raise ValueError(f"msgs has {msgs.shape[0]} rows for {B} images")
        with torch.cuda.device(eng.dev):
            mi = self._msgs_dev(msgs, eng.dev)
            on_dev = imgs.device == eng.dev
            preds_w = _result_buffer((B, cd, imgs.shape[-2], imgs.shape[-1]), torch.float32, imgs.device)

            def one(fr, oc, a, b):
                pw = preds_w[a:b] if on_dev else torch.empty(b - a, cd, fr.shape[-2], fr.shape[-1], device=eng.dev, dtype=torch.float32)
                self._embed_frames(eng, fr, mi[a:b], oc, step=1, video_mode=0, antialias=aa, lowres=lowres_attenuation, preds_w=pw)
                if not on_dev:
                    preds_w[a:b].copy_(pw, non_blocking=True)       # _run_chunks synchronises before it returns
            out = self._run_chunks(eng, imgs, max(1, int(getattr(self, "chunk_size", 32))), one)
        return {"msgs": msgs, "preds_w": preds_w, "imgs_w": out}

    @torch.no_grad()
    def detect(self, imgs: torch.Tensor, interpolation: dict = None) -> dict:
        """wam.py:206-234."""
        eng = self._engine()
        aa = _antialias_flag(interpolation)
        if imgs.shape[0] == 0:
            return {"preds": imgs.new_zeros((0, self.embedder.cfg.nbits + 1))}
        with torch.cuda.device(eng.dev):
            x = N.f32c(imgs.to(eng.dev))
            return {"preds": self._detect_frames(eng, x, (self.img_size, self.img_size), aa).to(imgs.device)}

    def _augment_detect(self, eng: HipEngine, imgs_w: torch.Tensor, imgs: torch.Tensor, masks, is_video: bool, aa: bool):
        """wam.py:115-125 / videoseal.py:231-243: augment -> resize to the processing size -> detector (device tensors)."""
        from . import augmentation as A
        imgs_aug, masks, selected = self.augmenter(imgs_w, imgs, masks.to(eng.dev) if torch.is_tensor(masks) else masks,
                                                   is_video=is_video, do_resize=False)
        S = (self.img_size, self.img_size)
        if tuple(imgs_aug.shape[-2:]) != S:
            imgs_aug = A.resize(imgs_aug, S, aa)
        rgb, _ = eng.resize_pre(N.f32c(imgs_aug), S, False, want_rgb=True, mul=2.0, add=-1.0, tag="det.in")
        return imgs_aug, masks, selected, eng.extractor_forward(rgb).clone()

    def _trainable(self) -> Tuple[bool, bool]:
        """(embedder, detector) have parameters that want gradients in the current autograd mode"""
        if not torch.is_grad_enabled():
            return False, False
        cfg = self.embedder.cfg
        emb = any(p.requires_grad for p in self.embedder.parameters())
        det = any(p.requires_grad for p in self.detector.parameters())
        return emb, det

    def _forward_graph(self, x: torch.Tensor, masks, mi: torch.Tensor, *, step: int, video_mode: int, aa: bool, lowres: bool, is_video: bool,
                       emb_train: bool):
        """the differentiable forward (wam.py:86-125 / videoseal.py:181-243): EmbedTrainFn -> augmenter nodes -> ResizeFn -> DetectTrainFn"""
        from . import augmentation as A
        from . import autograd as AG
        eng = self._engine()
        if emb_train:
            named = [(k, p) for k, p in AG._named_unique(self.embedder, "embedder.")]
            opts = dict(step=step, video_mode=video_mode, antialias=aa, lowres=lowres)
            imgs_w, preds_w = AG.EmbedTrainFn.apply(self, x, mi, opts, [k for k, _ in named], *[p for _, p in named])
        else:
            with torch.no_grad():
                imgs_w = torch.empty_like(x)
                preds_w = torch.empty(x.shape[0], self.embedder.cfg.out_ch, x.shape[-2], x.shape[-1], device=eng.dev, dtype=torch.float32)
                self._embed_frames(eng, x, mi, imgs_w, step=step, video_mode=video_mode, antialias=aa, lowres=lowres, preds_w=preds_w, fwd_order=True)
        imgs_aug, masks, selected = self.augmenter(imgs_w, x, masks.to(eng.dev) if torch.is_tensor(masks) else masks, is_video=is_video,
                                                   do_resize=False)
        S = (self.img_size, self.img_size)
        if tuple(imgs_aug.shape[-2:]) != S:
            imgs_aug = A.resize(imgs_aug, S, aa)
        named = AG._named_unique(self.detector, "detector.")
        if imgs_aug.requires_grad or any(p.requires_grad for _, p in named):
            preds = AG.DetectTrainFn.apply(self, imgs_aug, [k for k, _ in named], *[p for _, p in named])
        else:
            with torch.no_grad():
                rgb, _ = eng.resize_pre(N.f32c(imgs_aug), S, False, want_rgb=True, mul=2.0, add=-1.0, tag="det.in")
                preds = eng.extractor_forward(rgb).clone()
        return imgs_w, preds_w, imgs_aug, masks, selected, preds

    def forward(self, imgs: torch.Tensor, masks: torch.Tensor, msgs: torch.Tensor = None, interpolation: dict = None) -> dict:
        """wam.py:68-132: embed (blend, then full-resolution attenuation(imgs, imgs_w), clamp) -> augmenter -> resize to img_size ->
        detector.  `preds_w` is the UN-attenuated resized delta and `imgs_aug` the resized augmented batch, like the reference.
        BatchNorm follows ``self.embedder.training``.  With autograd enabled and trainable parameters the outputs carry a graph whose
        backward runs on the HIP kernels (autograd.py); otherwise values only."""
        emb_t, det_t = self._trainable()
        if not (emb_t or det_t):
            with torch.no_grad():
                return self._forward_values(imgs, masks, msgs, interpolation)
        if msgs is None:
            msgs = self.get_random_msg(imgs.shape[0]).to(imgs.device)
        eng = self._engine()
        aa = _antialias_flag(_DEFAULT_INTERP if interpolation is None else interpolation)
        back = imgs.device
        with torch.cuda.device(eng.dev):
            x = N.f32c(imgs.detach().to(eng.dev))
            imgs_w, preds_w, imgs_aug, masks, selected, preds = self._forward_graph(
                x, masks, self._msgs_dev(msgs, eng.dev), step=1, video_mode=0, aa=aa, lowres=False, is_video=False, emb_train=emb_t)
        to = (lambda t: t.to(back) if torch.is_tensor(t) else t)     # noqa: E731
        return {"msgs": msgs, "masks": to(masks), "preds_w": to(preds_w), "imgs_w": to(imgs_w), "imgs_aug": to(imgs_aug),
                "preds": to(preds), "selected_aug": selected}

    def _forward_values(self, imgs: torch.Tensor, masks: torch.Tensor, msgs: torch.Tensor = None, interpolation: dict = None) -> dict:
        if msgs is None:
            msgs = self.get_random_msg(imgs.shape[0]).to(imgs.device)
        eng = self._engine()
        aa = _antialias_flag(_DEFAULT_INTERP if interpolation is None else interpolation)
        back = imgs.device
        with torch.cuda.device(eng.dev):
            x = N.f32c(imgs.to(eng.dev))
            B = x.shape[0]
            out = torch.empty_like(x)
            preds_w = torch.empty(B, self.embedder.cfg.out_ch, x.shape[-2], x.shape[-1], device=eng.dev, dtype=torch.float32)
            self._embed_frames(eng, x, self._msgs_dev(msgs, eng.dev), out, step=1, video_mode=0, antialias=aa, lowres=False,
                               preds_w=preds_w, fwd_order=True)
            imgs_aug, masks, selected, preds = self._augment_detect(eng, out, x, masks, False, aa)
        to = (lambda t: _to_caller(t, back))     # noqa: E731
        return {"msgs": msgs, "masks": to(masks), "preds_w": to(preds_w), "imgs_w": to(out), "imgs_aug": to(imgs_aug),
                "preds": to(preds), "selected_aug": selected}


class Videoseal(Wam):
    """Video path (models/videoseal.py:15-428): key-frame stepping, chunking, frame aggregation."""

    def __init__(self, embedder, detector, augmenter, attenuation=None, scaling_w: float = 1.0, scaling_i: float = 1.0,
                 img_size: int = 256, clamp: bool = True, chunk_size: int = 8, step_size: int = 4,
                 blending_method: str = "additive", video_mode: str = "repeat", lowres_attenuation: bool = False) -> None:
        super().__init__(embedder, detector, augmenter, attenuation, scaling_w, scaling_i, clamp, img_size, blending_method)
        self.chunk_size, self.step_size = chunk_size, step_size
        if self.embedder.cfg.chunk_size != int(chunk_size) or self.embedder.cfg.step_size != int(step_size):
            import dataclasses            # (a fresh object: a card's cfg may be shared between models)
            self.embedder.cfg = self.detector.cfg = dataclasses.replace(self.embedder.cfg, chunk_size=int(chunk_size), step_size=int(step_size))
        self.video_mode = video_mode
        self.lowres_attenuation = lowres_attenuation

    def _embed_clip(self, imgs: torch.Tensor, msgs: torch.Tensor, interpolation, lowres_attenuation: bool) -> torch.Tensor:
        if self.video_mode not in N.VIDEO_MODES:
            raise ValueError(f"unknown video_mode {self.video_mode}")
        eng = self._engine()
        aa = _antialias_flag(interpolation)
        step, ck = int(self.step_size), int(self.chunk_size)
        with torch.cuda.device(eng.dev):
            mi = self._msgs_dev(msgs, eng.dev)
            vm = N.VIDEO_MODES[self.video_mode]
            return self._run_chunks(eng, imgs, ck * step,       # frames per chunk (videoseal.py:292-297)
                                    lambda fr, oc, a, b: self._embed_frames(eng, fr, mi, oc, step=step, video_mode=vm, antialias=aa,
                                                                            lowres=lowres_attenuation))

    @torch.no_grad()
    def embed_group(self, frames: torch.Tensor, msgs: torch.Tensor, chunk: int, interpolation: dict = None,
                    lowres_attenuation: bool = False, on_tail=None, tail_batch: int = 0) -> torch.Tensor:
        """`frames` (device-resident fp32 [F,3,H,W] or uint8 RGB24 [F,H,W,3]) = consecutive caller chunks of `chunk` frames (the last one may be
        short), `chunk` a multiple of step_size.  Returns what `torch.cat([embed(c, msgs, is_video=True)['imgs_w'] for c in chunks])` returns
        (embed_u8 for uint8), with the key frames of ALL chunks going through the U-Net as one batch: 16-frame streaming calls
        (inference_streaming.py:83-107) carry 4 key frames each, which leaves the matrix kernels at a third of the rate they reach with 32.
        The per-chunk semantics of videoseal.py:303-344 are kept by expanding the watermark chunk by chunk (`_embed_frames_eager`); values
        differ from the per-chunk calls only by the summation order of the dense layers (a K split is a function of the batch shape)."""
        if self.video_mode not in N.VIDEO_MODES:
            raise ValueError(f"unknown video_mode {self.video_mode}")
        assert msgs.shape[0] == 1, "Message should be unique"
        eng = self._engine()
        step = int(self.step_size)
        if chunk % step:
            raise ValueError(f"chunk ({chunk}) must be a multiple of step_size ({step}): the key frames of the group are every step-th frame")
        if chunk > int(self.chunk_size) * step:
            # embed() itself walks a call in chunk_size * step_size frames (videoseal.py:291-297) and the watermark expansion restarts there;
            # a caller chunk that spans several of them is not one expansion span
            raise ValueError(f"chunk ({chunk}) exceeds chunk_size * step_size = {int(self.chunk_size) * step}: embed() would split such a call "
                             f"internally; raise model.chunk_size or pass smaller chunks")
        if frames.device != eng.dev:
            raise ValueError("embed_group wants device-resident frames")
        u8 = frames.dtype == torch.uint8
        if u8 and not self.clamp:
            raise NotImplementedError("uint8 output needs clamp=True ((x * 255).byte() is undefined outside [0, 1])")
        aa = _antialias_flag(interpolation)
        with torch.cuda.device(eng.dev):
            src = frames.contiguous() if u8 else N.f32c(frames)
            out = torch.empty_like(src)
            if on_tail is not None and not self.use_graphs:
                # (streaming.py) `on_tail(first, imgs_w[first:last])` as soon as another `tail_batch` watermarked frames are issued
                self._embed_frames_eager(eng, src, self._msgs_dev(msgs, eng.dev), out, step=step, video_mode=N.VIDEO_MODES[self.video_mode],
                                         antialias=aa, lowres=lowres_attenuation, tail_span=int(chunk), tail_batch=int(tail_batch),
                                         on_tail=lambda a, b: on_tail(a, out[a:b]))
                return out
            self._embed_frames(eng, src, self._msgs_dev(msgs, eng.dev), out, step=step, video_mode=N.VIDEO_MODES[self.video_mode], antialias=aa,
                               lowres=lowres_attenuation, tail_span=int(chunk))
            if on_tail is not None:
                on_tail(0, out)
        return out

    @torch.no_grad()
    def embed(self, imgs: torch.Tensor, msgs: torch.Tensor = None, is_video: bool = True, interpolation: dict = None,
              lowres_attenuation: bool = False) -> dict:
        """videoseal.py:258-350."""
        if not is_video:
            return super().embed(imgs, msgs, interpolation, lowres_attenuation)
        if msgs is None:
            msgs = self.get_random_msg()
        else:
            assert msgs.shape[0] == 1, "Message should be unique"
        out = self._embed_clip(imgs, msgs, interpolation, lowres_attenuation)
        return {"imgs_w": out, "msgs": msgs[0:1].repeat(len(imgs), 1)}

    def _detect_clip(self, imgs: torch.Tensor, interpolation) -> torch.Tensor:
        eng = self._engine()
        aa = _antialias_flag(interpolation)
        S = (self.img_size, self.img_size)
        preds = []
        with torch.cuda.device(eng.dev):
            self._run_chunks(eng, imgs, max(1, int(self.chunk_size)),
                             lambda fr, oc, a, b: preds.append(self._detect_frames(eng, fr, S, aa)), want_out=False, extra=preds.clear)
            return torch.cat(preds, dim=0).to(imgs.device)

    @torch.no_grad()
    def detect(self, imgs: torch.Tensor, is_video: bool = True, interpolation: dict = None) -> dict:
        """videoseal.py:352-388."""
        if not is_video:
            return super().detect(imgs) if interpolation is None else super().detect(imgs, interpolation)
        if imgs.shape[0] == 0:
            self._engine()
            return {"preds": imgs.new_zeros((0, self.embedder.cfg.nbits + 1))}
        return {"preds": self._detect_clip(imgs, interpolation)}

    # ---- uint8 RGB24 clips, the data format on either side of the path in inference_streaming.py
    @torch.no_grad()
    def embed_u8(self, clip: torch.Tensor, msgs: torch.Tensor = None, interpolation: dict = None,
                 lowres_attenuation: bool = True) -> dict:
        """inference_streaming.py:23-32 (`embed_video_clip`) in one pass: clip uint8 [F,H,W,3] (RGB24, as read from the ffmpeg
        pipe) -> {'imgs_w': uint8 [F,H,W,3], 'msgs'}.  Equals (embed(clip.float().permute(0,3,1,2)/255, is_video=True,
        lowres_attenuation=...)['imgs_w'] * 255).byte().permute(0,2,3,1) with the conversions fused into the resize and tail
        kernels (3 B/pixel instead of 12 B/pixel through HBM, no fp32 copy of the clip)."""
        if clip.dtype != torch.uint8 or clip.dim() != 4 or clip.shape[-1] != 3:
            raise ValueError("embed_u8 wants a uint8 RGB24 clip [F, H, W, 3]")
        if not self.clamp:
            raise NotImplementedError("uint8 output needs clamp=True ((x * 255).byte() is undefined outside [0, 1])")
        if msgs is None:
            msgs = self.get_random_msg()
        else:
            assert msgs.shape[0] == 1, "Message should be unique"
        out = self._embed_clip(clip.contiguous(), msgs, interpolation, lowres_attenuation)
        return {"imgs_w": out, "msgs": msgs[0:1].repeat(len(clip), 1)}

    @torch.no_grad()
    def detect_u8(self, clip: torch.Tensor, interpolation: dict = None) -> dict:
        """inference_streaming.py:119-125 (`detect_video_clip`): uint8 [F,H,W,3] -> {'preds': [F, 1+nbits]}."""
        if clip.dtype != torch.uint8 or clip.dim() != 4 or clip.shape[-1] != 3:
            raise ValueError("detect_u8 wants a uint8 RGB24 clip [F, H, W, 3]")
        if clip.shape[0] == 0:
            self._engine()
            return {"preds": torch.zeros((0, self.embedder.cfg.nbits + 1), device=clip.device)}
        return {"preds": self._detect_clip(clip.contiguous(), interpolation)}

    def extract_message(self, imgs: torch.Tensor, aggregation: str = "avg",
                        interpolation: dict = {"mode": "bilinear", "align_corners": False, "antialias": False}) -> torch.Tensor:
        """videoseal.py:390-428."""
        preds = self.detect(imgs, is_video=True, interpolation=interpolation)["preds"]
        bit_preds = preds[:, 1:]
        decoded = aggregate_bits(bit_preds, aggregation)
        return (decoded > 0).squeeze().unsqueeze(0)

    def forward(self, imgs: torch.Tensor, masks: torch.Tensor, msgs: torch.Tensor = None, is_video: bool = True):
        """videoseal.py:120-161 (differentiable, see Wam.forward)."""
        assert not (is_video and len(imgs.shape) not in [4, 5]), \
            "If is_video is True, input shape should be [b, frames, c, h, w] or [frames, c, h, w]"
        assert not (not is_video and len(imgs.shape) != 4), "If is_video is False, input shape should be [b, c, h, w]"
        if not is_video:
            return super().forward(imgs, masks, msgs)
        if imgs.dim() == 5:
            return [self.video_forward(imgs[i], masks[i] if masks is not None else None, msgs[i] if msgs is not None else None)
                    for i in range(imgs.shape[0])]
        return self.video_forward(imgs, masks, msgs)

    def video_forward(self, imgs, masks, msgs=None, interpolation: dict = None) -> dict:
        """videoseal.py:163-256: the whole clip as one chunk, key frames every step_size, video_mode expansion, attenuation at low
        resolution (self.lowres_attenuation) or as attenuation(imgs, imgs_w) at full resolution, clamp, augment, resize, detect."""
        if msgs is None:
            msgs = self.get_random_msg()
        else:
            assert msgs.shape[0] == 1, "Message should be unique"
        msgs = msgs.to(imgs.device)
        if self.video_mode not in N.VIDEO_MODES:
            raise ValueError(f"unknown video_mode {self.video_mode}")
        eng = self._engine()
        aa = _antialias_flag(_DEFAULT_INTERP if interpolation is None else interpolation)
        back = imgs.device
        emb_t, det_t = self._trainable()
        with torch.cuda.device(eng.dev), torch.set_grad_enabled(emb_t or det_t):
            x = N.f32c(imgs.detach().to(eng.dev))
            if emb_t or det_t:
                out, _, imgs_aug, masks, selected, preds = self._forward_graph(
                    x, masks, self._msgs_dev(msgs, eng.dev), step=int(self.step_size), video_mode=N.VIDEO_MODES[self.video_mode], aa=aa,
                    lowres=bool(self.lowres_attenuation), is_video=True, emb_train=emb_t)
            else:
                out = torch.empty_like(x)
                self._embed_frames(eng, x, self._msgs_dev(msgs, eng.dev), out, step=int(self.step_size),
                                   video_mode=N.VIDEO_MODES[self.video_mode], antialias=aa, lowres=bool(self.lowres_attenuation), fwd_order=True)
                imgs_aug, masks, selected, preds = self._augment_detect(eng, out, x, masks, True, aa)
        to = (lambda t: (t.to(back) if torch.is_tensor(t) else t) if (emb_t or det_t) else _to_caller(t, back))     # noqa: E731
        return {"msgs": msgs.expand(imgs.shape[0], -1), "masks": to(masks), "imgs_w": to(out), "imgs_aug": to(imgs_aug), "preds": to(preds),
                "selected_aug": selected}


def aggregate_bits(bit_preds: torch.Tensor, aggregation: Optional[str]) -> torch.Tensor:
    """Frame aggregation of videoseal.py:411-426 (tiny [F,k] reduction, plain torch on the caller's device)."""
    if aggregation is None:
        return bit_preds
    if aggregation == "avg":
        return bit_preds.mean(dim=0)
    if aggregation == "squared_avg":
        return (bit_preds * bit_preds.abs()).mean(dim=0)
    if aggregation == "l1norm_avg":
        return (bit_preds * torch.norm(bit_preds, p=1, dim=1).unsqueeze(1)).mean(dim=0)
    if aggregation == "l2norm_avg":
        return (bit_preds * torch.norm(bit_preds, p=2, dim=1).unsqueeze(1)).mean(dim=0)
    raise ValueError(f"unknown aggregation {aggregation}")


def build_model(cfg: ModelCfg, seed: int = 0) -> Videoseal:
    """cfg.py:120-144: embedder + extractor + identity augmenter + JND -> Videoseal (train mode, CPU, like the reference)."""
    return Videoseal(Embedder(cfg, seed), Extractor(cfg, seed), get_dummy_augmenter(),
                     attenuation=(JND(in_channels=cfg.jnd_in, out_channels=cfg.jnd_out) if cfg.jnd_in > 0 else None), scaling_w=cfg.scaling_w,
                     scaling_i=cfg.scaling_i, img_size=cfg.img_size, chunk_size=cfg.chunk_size, step_size=cfg.step_size,
                     blending_method=cfg.blending_method)
