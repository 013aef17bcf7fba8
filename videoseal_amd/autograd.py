"""torch.autograd integration of the HIP path: what lets the reference's UNMODIFIED training loop run on this module
(train.py:626-643:  outputs = wam(imgs, masks) -> VideosealLoss(...) -> loss.backward();  optimizer over embedder + extractor parameters,
train.py:330;  `get_last_layer()` probed with torch.autograd.grad, videosealloss.py:86-90;  DDP's gradient hooks, train.py:442-446).

Every node's forward AND backward is a launch sequence of hand-written gfx950 kernels (csrc/bwd_ops.hip, bwd_unet.hip, bwd_shell.hip);
autograd only carries the graph.  Nodes:
  EmbedTrainFn    frames + message + embedder parameters -> imgs_w, preds_w      (wam.py:86-113 / videoseal.py:181-228)
  DetectTrainFn   frames at the working size + detector parameters -> logits     (extractor.py:154-167)
  CropFlipFn, ResizeFn, MaskBlendFn, ColorFn, SteFn                              the augmentations between the two (augmenter.py:154-194)
  PercepLossFn, DecodeLossFn                                                     the generator-side loss terms (videosealloss.py:121-156)
Parameters enter the two network nodes as explicit inputs, so `.grad` accumulation, `torch.autograd.grad(loss, last_layer)` and DDP hooks work as
with any nn.Module.  The operands the backward needs live in the engine's workspace (tagged "tr."), owned by the LAST training forward of the
model: a backward through an older graph raises instead of reading overwritten buffers.  No CPU path: without the library or a GPU this raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import native as N


def needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else N.f32c(t)


# ----------------------------------------------------------------------------------------------------------------- augmentation nodes
class CropFlipFn(torch.autograd.Function):
    """geometric.py:94-124, 186-196 (vs_aug_crop_flip / vs_aug_crop_flip_bwd)"""

    @staticmethod
    def forward(ctx, x, i, j, h, w, flip):
        from . import augmentation as A
        ctx.geom = (x.shape, int(i), int(j), int(h), int(w), bool(flip))
        return A.crop_flip(x, i, j, h, w, flip)

    @staticmethod
    def backward(ctx, dy):
        shape, i, j, h, w, flip = ctx.geom
        dy = _c(dy)
        dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
        N.check(N.lib().vs_aug_crop_flip_bwd(N.ptr(dy), N.ptr(dx), shape[0] * shape[1], shape[-2], shape[-1], i, j, h, w, int(flip), N.stream()),
                "vs_aug_crop_flip_bwd")
        return dx, None, None, None, None, None


def resize_bwd(dy: torch.Tensor, in_size: Tuple[int, int], antialias: bool) -> torch.Tensor:
    """transpose of augmentation.resize / of the up-resize inside vs_embed_tail: dy [.., oh, ow] -> dx [.., H, W]"""
    dy = _c(dy)
    planes, oh, ow = dy.shape[0] * dy.shape[1], dy.shape[-2], dy.shape[-1]
    H, W = in_size
    dx = torch.empty(dy.shape[0], dy.shape[1], H, W, device=dy.device, dtype=torch.float32)
    tmp = torch.empty(planes * oh * W, device=dy.device, dtype=torch.float32)
    N.check(N.lib().vs_resize_nchw_bwd(N.ptr(dy), N.ptr(dx), planes, H, W, oh, ow, int(antialias), N.ptr(tmp), N.stream()), "vs_resize_nchw_bwd")
    return dx


class ResizeFn(torch.autograd.Function):
    """F.interpolate(bilinear, antialias) / torchvision resize (wam.py:117, geometric.py:62-91): vs_resize_nchw and its transpose"""

    @staticmethod
    def forward(ctx, x, size, antialias):
        from . import augmentation as A
        ctx.geom = (tuple(x.shape[-2:]), bool(antialias))
        return A.resize(x, size, antialias)

    @staticmethod
    def backward(ctx, dy):
        in_size, aa = ctx.geom
        return resize_bwd(dy, in_size, aa), None, None


class MaskBlendFn(torch.autograd.Function):
    """augmenter.py:175  imgs_w * m + imgs * (1 - m): d imgs_w = m * dy, d imgs = (1 - m) * dy"""

    @staticmethod
    def forward(ctx, imgs_w, imgs, mask):
        from . import augmentation as A
        ctx.save_for_backward(mask)
        return A.mask_blend(imgs_w, imgs, mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy, m = _c(dy), _c(mask)
        F_, Cc, H, W = dy.shape
        outs = []
        for k in range(2):
            if ctx.needs_input_grad[k]:
                dx = torch.empty_like(dy)
                N.check(N.lib().vs_mask_mul(N.ptr(dy), N.ptr(m), N.ptr(dx), F_, Cc, H, W, k, N.stream()), "vs_mask_mul")
                outs.append(dx)
            else:
                outs.append(None)
        return outs[0], outs[1], None


class ColorFn(torch.autograd.Function):
    """valuemetric.py:53-175: brightness / contrast / saturation (torchvision's blend + clamp), grayscale, and hue (the chain rule through
    torchvision's RGB -> HSV -> RGB round trip with autograd's conventions, bwd_shell.hip::hue_bwd)"""

    @staticmethod
    def forward(ctx, x, op, factor):
        from . import augmentation as A
        x = A._dev(x)
        F_, Cc, H, W = x.shape
        L = N.lib()
        out = torch.empty_like(x)
        scratch = torch.empty(int(L.vs_aug_color_scratch_floats(F_, H, W)), device=x.device, dtype=torch.float32)
        N.check(L.vs_aug_color(N.ptr(x), N.ptr(out), F_, H, W, A.COLOR_OPS[op], float(factor), N.ptr(scratch), N.stream()), "vs_aug_color")
        ctx.save_for_backward(x, scratch[-F_:].clone() if op == "contrast" else x.new_empty(0))     # per-frame gray means of the forward pass
        ctx.op, ctx.factor = A.COLOR_OPS[op], float(factor)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, means = ctx.saved_tensors
        dy = _c(dy)
        F_, _, H, W = x.shape
        L = N.lib()
        dx = torch.empty_like(x)
        scratch = torch.empty(int(L.vs_aug_color_bwd_scratch_floats(F_, H, W)) + 2, device=x.device, dtype=torch.float64).view(torch.float32)
        N.check(L.vs_aug_color_bwd(N.ptr(x), N.ptr(dy), N.ptr(dx), F_, H, W, ctx.op, ctx.factor, N.ptr(means) if means.numel() else None,
                                   N.ptr(scratch), N.stream()), "vs_aug_color_bwd")
        return dx, None, None


class BlurFn(torch.autograd.Function):
    """valuemetric.py:108-128 GaussianBlur (torchvision gaussian_blur, reflection padding): forward kernel, adjoint = the same separable filter
    over the reflection-padded gradient folded back (vs_gaussian_blur_bwd)"""

    @staticmethod
    def forward(ctx, x, kernel_size):
        from . import augmentation as A
        with torch.no_grad():
            y = A.gaussian_blur(x.detach(), kernel_size)
        ctx.k, ctx.shape = int(kernel_size), tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        k = ctx.k
        planes, H, W = ctx.shape[0] * ctx.shape[1], ctx.shape[-2], ctx.shape[-1]
        sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
        tmp, dx = torch.empty_like(dy), torch.empty_like(dy)
        N.check(N.lib().vs_gaussian_blur_bwd(N.ptr(dy), N.ptr(tmp), N.ptr(dx), planes, H, W, k, sigma, N.stream()), "vs_gaussian_blur_bwd")
        return dx, None


class WarpFn(torch.autograd.Function):
    """geometric.py:28-59 Rotate (nearest) and :127-183 Perspective (bilinear) on vs_aug_warp; adjoint in gather form (vs_aug_warp_bwd).
    coeffs: the kernel's sampling coefficients; out_hw: output size; the 3 x 3 inverse that bounds the adjoint's search is computed here in fp64."""

    @staticmethod
    def forward(ctx, x, kind, coeffs, bilinear, out_hw):
        import ctypes as C
        import numpy as np
        from . import augmentation as A
        x = A._dev(x)
        planes, H, W = x.shape[0] * x.shape[1], x.shape[-2], x.shape[-1]
        oh, ow = out_hw
        t = [float(v) for v in coeffs]
        carr = (C.c_float * len(t))(*t)
        out = torch.empty(x.shape[0], x.shape[1], oh, ow, device=x.device, dtype=torch.float32)
        N.check(N.lib().vs_aug_warp(N.ptr(x), N.ptr(out), planes, H, W, oh, ow, int(kind), carr, int(bilinear), N.stream()), "vs_aug_warp")
        # M: (cx, cy, 1) of an output pixel centre -> homogeneous input pixel-centre coordinates (ix + 0.5, iy + 0.5, 1) * w  (aug.hip::warp_kernel)
        if kind == 0:
            M = np.array([[0.5 * W * t[0], 0.5 * W * t[1], 0.5 * W * (t[2] + 1.0 - 0.5 * ow * t[0] - 0.5 * oh * t[1])],
                          [0.5 * H * t[3], 0.5 * H * t[4], 0.5 * H * (t[5] + 1.0 - 0.5 * ow * t[3] - 0.5 * oh * t[4])],
                          [0.0, 0.0, 1.0]], dtype=np.float64)
        else:
            M = np.array([[W / ow * t[0], W / ow * t[1], W / ow * t[2]], [H / oh * t[3], H / oh * t[4], H / oh * t[5]], [t[6], t[7], 1.0]],
                         dtype=np.float64)
        inv = np.linalg.inv(M)
        inv = inv / inv[2, 2] if abs(inv[2, 2]) > 1e-12 else inv
        ctx.args = (planes, H, W, oh, ow, int(kind), t, int(bilinear), [float(v) for v in inv.reshape(-1)])
        return out

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        planes, H, W, oh, ow, kind, t, bilinear, inv = ctx.args
        dy = _c(dy)
        dx = torch.empty(dy.shape[0], dy.shape[1], H, W, device=dy.device, dtype=torch.float32)
        N.check(N.lib().vs_aug_warp_bwd(N.ptr(dy), N.ptr(dx), planes, H, W, oh, ow, kind, (C.c_float * len(t))(*t), bilinear, (C.c_float * 9)(*inv),
                                        N.stream()), "vs_aug_warp_bwd")
        return dx, None, None, None, None


class GatherFramesFn(torch.autograd.Function):
    """whole-frame gathers of augmentation/video.py (DropFrame :491-529, SpeedChange :283-313, TemporalReorder :319-405): out[o] = frames[idx[o]];
    adjoint: every source frame receives the sum of the outputs that copied it (lists built on the host, summed in ascending output order)"""

    @staticmethod
    def forward(ctx, x, indices):
        from . import augmentation as A
        with torch.no_grad():
            y = A.gather_frames(x.detach(), indices)
        idx = [int(i) for i in torch.as_tensor(indices).reshape(-1).tolist()]
        ctx.idx, ctx.shape = idx, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        n_src = ctx.shape[0]
        lists = [[] for _ in range(n_src)]
        for o, f in enumerate(ctx.idx):
            lists[f].append(o)
        start, outs = [0], []
        for l in lists:
            outs.extend(l)
            start.append(len(outs))
        dev = dy.device
        st_t = torch.tensor(start, dtype=torch.int32, device=dev)
        ou_t = torch.tensor(outs if outs else [0], dtype=torch.int32, device=dev)
        dx = torch.empty(ctx.shape, device=dev, dtype=torch.float32)
        fsz = dx[0].numel()
        N.check(N.lib().vs_aug_gather_frames_bwd(N.ptr(dy), N.ptr(st_t), N.ptr(ou_t), N.ptr(dx), n_src, fsz, N.stream()), "vs_aug_gather_frames_bwd")
        return dx, None


class WindowAverageFn(torch.autograd.Function):
    """augmentation/video.py:411-486 WindowAveraging: forward kernel, exact adjoint (vs_aug_window_average_bwd)"""

    @staticmethod
    def forward(ctx, x, window_size, alpha):
        from . import augmentation as A
        with torch.no_grad():
            y = A.window_average(x.detach(), window_size, alpha)
        ctx.hw, ctx.alpha = int(window_size) // 2, float(alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        N.check(N.lib().vs_aug_window_average_bwd(N.ptr(dy), N.ptr(dx), dy.shape[0], dy[0].numel(), ctx.hw, ctx.alpha, N.stream()),
                "vs_aug_window_average_bwd")
        return dx, None, None


class SteFn(torch.autograd.Function):
    """straight-through estimator `x + (op(x) - x).detach()` of JPEG / MedianFilter / the video codecs (valuemetric.py:35, 90; video.py:113):
    forward value op(x), identity gradient.  clamp01: JPEG.forward clamps to [0, 1] in front of the estimator (valuemetric.py:41), which
    masks the gradient outside."""

    @staticmethod
    def forward(ctx, x, y, clamp01):
        ctx.clamp01 = bool(clamp01)
        if clamp01:
            ctx.save_for_backward(x)
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        if not ctx.clamp01:
            return dy, None, None
        (x,) = ctx.saved_tensors
        dy, x = _c(dy), _c(x)
        dx = torch.empty_like(dy)
        N.check(N.lib().vs_clamp01_bwd(N.ptr(x), N.ptr(dy), N.ptr(dx), dy.numel(), N.stream()), "vs_clamp01_bwd")
        return dx, None, None


class NoAdjointFn(torch.autograd.Function):
    """an op that has no derivative in the reference either (`passthrough=False` JPEG / MedianFilter): the forward value is exact, the backward
    raises instead of silently cutting the graph between the decoding loss and the embedder"""

    @staticmethod
    def forward(ctx, x, y, what):
        ctx.what = what
        return y.view_as(y)

    @staticmethod
    def backward(ctx, dy):
        raise NotImplementedError(f"{ctx.what}: no adjoint kernel for this augmentation in the HIP path -- remove it from the training augmentation "
                                  f"set or detach its input")


# ----------------------------------------------------------------------------------------------------------------- network nodes
def _named_unique(module: torch.nn.Module, prefix: str) -> List[Tuple[str, torch.nn.Parameter]]:
    """(state_dict name, parameter) once per tensor -- the message table is registered twice (embedder.py:141-142 + unet.py:128)"""
    seen, out = set(), []
    for k, p in module.named_parameters(remove_duplicate=False):
        if id(p) not in seen:
            seen.add(id(p))
            out.append((prefix + k, p))
    return out


_ALIASES = {"embedder.msg_processor.msg_embeddings.weight": "embedder.unet.msg_processor.msg_embeddings.weight"}


def _check_generation(model, what: str, gen: int) -> None:
    if model._train_gen.get(what) != gen:
        raise RuntimeError(f"backward through a stale {what} graph: the operands of the HIP backward live in the engine's workspace and belong to "
                           f"the model's most recent training forward (run backward before the next forward of the same model)")


class EmbedTrainFn(torch.autograd.Function):
    """frames x [F, 3, H, W] in [0, 1] (no gradient), msgs int32 -> imgs_w, preds_w.  opts: step, video_mode, antialias, lowres"""

    @staticmethod
    def forward(ctx, model, x, msgs_i32, opts, names, *params):
        from .training import EmbedderBackward
        eng = model._engine()
        if x.requires_grad:
            raise NotImplementedError("gradients with respect to the input frames are not built (the training loop does not need them)")
        if model._emb_bwd is None:
            model._emb_bwd = EmbedderBackward(model)
        eb = model._emb_bwd
        eng.begin_training_pass()
        S = (model.img_size, model.img_size)
        step, vm, aa, lowres = opts["step"], opts["video_mode"], opts["antialias"], opts["lowres"]
        att = model.attenuation is not None
        rgb, key = eng.resize_pre(x, S, aa, want_rgb=(att and lowres), want_key=True, key_step=step, tag="tr.rs")
        delta, saved = eb.forward_keep(eng, key, msgs_i32, update_running=True)
        hmap_low = eng.jnd_lowres(rgb).clone() if (att and lowres) else None
        out = torch.empty_like(x)
        Cd = eng.cfg.out_ch
        preds_w = torch.empty(x.shape[0], Cd, x.shape[-2], x.shape[-1], device=x.device, dtype=torch.float32)
        eng.embed_tail(x, out, delta, step=step, video_mode=vm, hmap_low=hmap_low, attenuate=(2 if (att and not lowres) else int(att)),
                       clamp=model.clamp, antialias=aa, scaling_i=model.blender.scaling_i, scaling_w=model.blender.scaling_w, preds_w=preds_w)
        model._train_gen["embedder"] = gen = model._train_gen.get("embedder", 0) + 1
        ctx.model, ctx.saved, ctx.hmap_low, ctx.gen, ctx.names = model, saved, hmap_low, gen, names
        model._last_train_saved = saved          # (tests read the ReLU masks of this forward from it)
        ctx.flags = (step, vm, aa, lowres, att, bool(model.clamp), float(model.blender.scaling_i), float(model.blender.scaling_w), delta.shape[0])
        ctx.shapes = [p.shape for p in params]
        ctx.save_for_backward(x, preds_w)
        ctx.set_materialize_grads(False)
        return out, preds_w

    @staticmethod
    def backward(ctx, d_out, d_preds):
        model = ctx.model
        if d_out is None and d_preds is None:
            return (None,) * (5 + len(ctx.names))
        _check_generation(model, "embedder", ctx.gen)
        x, preds_w = ctx.saved_tensors
        eng, eb, L, st = model._engine(), model._emb_bwd, N.lib(), N.stream()
        step, vm, aa, lowres, att, clamp, si, sw, nkey = ctx.flags
        F_, _, H, W = x.shape
        Cd, S = eng.cfg.out_ch, model.img_size
        with torch.cuda.device(eng.dev):
            hm_full = eng.jnd_full(x) if (att and not lowres) else None
            g_full = eng.buf("tr.tail.gfull", F_ * Cd * H * W)
            N.check(L.vs_embed_tail_bwd(N.ptr(x), N.ptr(preds_w), N.ptr(hm_full), N.ptr(_c(d_out)), N.ptr(_c(d_preds)), F_, H, W, Cd, int(clamp), si, sw,
                                        N.ptr(g_full), st), "vs_embed_tail_bwd")
            if (H, W) != (S, S):
                g_low = eng.buf("tr.tail.glow", F_ * Cd * S * S)
                tmp = eng.buf("tr.tail.tmp", F_ * Cd * H * S)
                N.check(L.vs_resize_nchw_bwd(N.ptr(g_full), N.ptr(g_low), F_ * Cd, S, S, H, W, int(aa), N.ptr(tmp), st), "vs_resize_nchw_bwd")
            else:
                g_low = g_full
            d_delta = eng.buf("tr.tail.ddelta", nkey * Cd * S * S)
            N.check(L.vs_tail_key_reduce(N.ptr(g_low), N.ptr(ctx.hmap_low), F_, Cd, S, S, step, vm, nkey, N.ptr(d_delta), st), "vs_tail_key_reduce")
            want = [n for n, need in zip(ctx.names, ctx.needs_input_grad[5:]) if need]
            last = "embedder.unet.outc."
            G = eb.backward(eng, ctx.saved, d_delta.view(nkey, Cd, S, S), outc_only=all(n.startswith(last) for n in want))
        grads = []
        for n, need, shp in zip(ctx.names, ctx.needs_input_grad[5:], ctx.shapes):
            g = G.get(_ALIASES.get(n, n)) if need else None
            if need and g is None:
                raise N.NativeError(f"no gradient produced for {n}")
            grads.append(g.reshape(shp) if g is not None else None)
        return (None, None, None, None, None) + tuple(grads)


class DetectTrainFn(torch.autograd.Function):
    """frames at the working size [B, 3, S, S] in [0, 1] + detector parameters -> logits [B, 1 + nbits]"""

    @staticmethod
    def forward(ctx, model, x, names, *params):
        from .training import DetectorStep
        eng = model._engine()
        if model._det_bwd is None:
            model._det_bwd = DetectorStep(model)
        ds = model._det_bwd
        eng.begin_training_pass()
        x = N.f32c(x)
        rgb, _ = eng.resize_pre(x, (x.shape[-2], x.shape[-1]), False, want_rgb=True, mul=2.0, add=-1.0, tag="tr.det.in")
        logits, saved = ds._forward(eng, rgb)
        model._train_gen["detector"] = gen = model._train_gen.get("detector", 0) + 1
        ctx.model, ctx.saved, ctx.gen, ctx.names, ctx.shapes = model, saved, gen, names, [p.shape for p in params]
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        _check_generation(model, "detector", ctx.gen)
        eng, ds = model._engine(), model._det_bwd
        want_params = any(ctx.needs_input_grad[3:])
        want_input = ctx.needs_input_grad[1]
        with torch.cuda.device(eng.dev):
            res = ds._backward(eng, ctx.saved, _c(dlogits), want_params=want_params, want_input=want_input)
        G, dimg = res if want_input else (res, None)
        grads = []
        for n, need, shp in zip(ctx.names, ctx.needs_input_grad[3:], ctx.shapes):
            g = G.get(n) if need else None
            if need and g is None:
                raise N.NativeError(f"no gradient produced for {n}")
            grads.append(g.reshape(shp) if g is not None else None)
        return (None, dimg, None) + tuple(grads)


# ----------------------------------------------------------------------------------------------------------------- loss nodes
class PercepLossFn(torch.autograd.Function):
    """losses/perceptual.py:20-28 'mse' / 'yuv' (yuvloss.py:11-27): mean((T (imgs_w - imgs))^2); gradient with respect to imgs_w"""

    @staticmethod
    def forward(ctx, imgs, imgs_w, yuv):
        a, b = N.f32c(imgs), N.f32c(imgs_w)
        F_, Cc, H, W = a.shape
        if Cc != 3 or b.shape != a.shape:
            raise ValueError("perceptual term: [F, 3, H, W] frames of equal shape")
        L = N.lib()
        part = torch.empty(int(L.vs_percep_partial_doubles(F_, H, W)), device=a.device, dtype=torch.float64)
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        N.check(L.vs_percep_mse(N.ptr(a), N.ptr(b), F_, H, W, int(yuv), N.ptr(part), N.ptr(loss), N.stream()), "vs_percep_mse")
        ctx.save_for_backward(a, b)
        ctx.yuv = int(yuv)
        return loss[0]

    @staticmethod
    def backward(ctx, up):
        a, b = ctx.saved_tensors
        F_, _, H, W = a.shape
        d = torch.empty_like(b)
        N.check(N.lib().vs_percep_mse_grad(N.ptr(a), N.ptr(b), F_, H, W, ctx.yuv, 1.0, N.ptr(d), N.stream()), "vs_percep_mse_grad")
        return None, d.mul_(up), None           # upstream gradient applied on the device (float(up) was a host synchronisation per backward)


class DecodeLossFn(torch.autograd.Function):
    """videosealloss.py:150-156: BCE-with-logits of preds[:, 1:] / temperature against the message bits, mean over everything (vs_bce_logits)"""

    @staticmethod
    def forward(ctx, preds, msgs_i32, temperature):
        p = N.f32c(preds)
        B, k = p.shape[0], p.shape[1] - 1
        if msgs_i32.dim() != 2 or msgs_i32.shape[1] != k or msgs_i32.shape[0] not in (1, B):
            raise ValueError(f"msgs must be [{B} or 1, {k}]")
        d = torch.empty_like(p)
        loss = torch.empty(1, device=p.device, dtype=torch.float32)
        N.check(N.lib().vs_bce_logits(N.ptr(p), N.ptr(msgs_i32), msgs_i32.shape[0], B, k, float(temperature), 1.0, N.ptr(d), N.ptr(loss), N.stream()),
                "vs_bce_logits")
        ctx.save_for_backward(d)
        return loss[0]

    @staticmethod
    def backward(ctx, up):
        (d,) = ctx.saved_tensors
        return d * up, None, None


def percep_loss(imgs: torch.Tensor, imgs_w: torch.Tensor, kind: str = "mse") -> torch.Tensor:
    if kind not in ("mse", "yuv"):
        raise NotImplementedError(f"perceptual loss {kind!r}: the HIP path has 'mse' and 'yuv' (the others need pretrained networks)")
    return PercepLossFn.apply(imgs, imgs_w, kind == "yuv")


def decoding_loss(preds: torch.Tensor, msgs: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    if preds.dim() != 2:
        raise NotImplementedError("per-pixel message predictions (videosealloss.py:157-169) are not produced by the ConvNeXt / ViT extractors")
    m = msgs.to(preds.device)
    if m.is_floating_point():
        m = m > 0.5
    return DecodeLossFn.apply(preds, m.to(torch.int32).contiguous(), temperature)
