"""ctypes binding of the MODEL-LEVEL C-ABI (include/videoseal_hip.h: vs_model_create / vs_model_embed / vs_model_detect).

This is the stub a non-Python host would write in its own FFI: card numbers + the reference state_dict go in once, whole-path
calls take raw device pointers, a workspace and a stream.  The Python `Videoseal` class (model.py) drives the operator-level
entry points itself (per-shape tile tuning, hipGraph capture); `CModel` exists to exercise and document the C boundary."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import native as N
from .layout import ModelCfg


class ModelCfgC(C.Structure):
    """mirror of vs_model_cfg_t"""
    _fields_ = [("nbits", C.c_int32), ("hidden", C.c_int32), ("img_size", C.c_int32), ("in_ch", C.c_int32), ("out_ch", C.c_int32),
                ("yuv", C.c_int32), ("nlev", C.c_int32), ("zc", C.c_int32 * 8), ("num_blocks", C.c_int32), ("last_tanh", C.c_int32),
                ("depths", C.c_int32 * 4), ("dims", C.c_int32 * 4), ("stem_stride", C.c_int32), ("attenuate", C.c_int32),
                ("clamp", C.c_int32), ("scaling_w", C.c_float), ("scaling_i", C.c_float), ("arith", C.c_int32), ("reserved_", C.c_int32)]


class TensorC(C.Structure):
    """mirror of vs_tensor_t"""
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


def _bind(L):
    if getattr(L, "_model_api_bound", False):
        return
    P, I, I64 = C.c_void_p, C.c_int, C.c_int64
    L.vs_model_create.argtypes = [C.POINTER(ModelCfgC), C.POINTER(TensorC), I, C.POINTER(P)]
    L.vs_model_create.restype = I
    L.vs_model_destroy.argtypes = [P]
    L.vs_model_destroy.restype = None
    L.vs_model_workspace_bytes.argtypes = [P, I, I, I, I]
    L.vs_model_workspace_bytes.restype = I64
    L.vs_model_embed.argtypes = [P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, I64, P]
    L.vs_model_embed.restype = I
    L.vs_model_detect.argtypes = [P, P, I, I, I, I, I, P, P, I64, P]
    L.vs_model_detect.restype = I
    L._model_api_bound = True


class CModel:
    """vs_model_t built from a ModelCfg and a reference-format state_dict (any device; copied to host fp32 for the call)."""

    def __init__(self, cfg: ModelCfg, state_dict: Dict[str, torch.Tensor], attenuate: bool = True, clamp: bool = True,
                 scaling_w: Optional[float] = None, scaling_i: Optional[float] = None):
        L = N.lib()
        _bind(L)
        if cfg.extractor != "convnext" or cfg.unet_norm != "batch":
            raise NotImplementedError("the model-level C-ABI covers the released cards (BatchNorm/ReLU U-Net + ConvNeXt-V2 extractor); "
                                      "the legacy videoseal_0.0 family runs through the operator-level entry points (videoseal_amd.engine)")
        self.cfg = cfg
        c = ModelCfgC()
        c.nbits, c.hidden, c.img_size, c.in_ch, c.out_ch, c.yuv = cfg.nbits, cfg.hidden, cfg.img_size, cfg.in_ch, cfg.out_ch, int(cfg.yuv)
        zc = cfg.zc
        c.nlev = len(zc) - 1
        for i, v in enumerate(zc):
            c.zc[i] = v
        c.num_blocks, c.last_tanh, c.stem_stride = cfg.num_blocks, int(cfg.last_tanh), cfg.stem_stride
        for i in range(4):
            c.depths[i], c.dims[i] = cfg.depths[i], cfg.dims[i]
        c.attenuate, c.clamp = int(attenuate), int(clamp)
        c.scaling_w = cfg.scaling_w if scaling_w is None else scaling_w
        c.scaling_i = cfg.scaling_i if scaling_i is None else scaling_i
        keep = []
        items = [(k, v) for k, v in state_dict.items() if v.dtype.is_floating_point]
        arr = (TensorC * len(items))()
        for i, (k, v) in enumerate(items):
            t = v.detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            arr[i].name, arr[i].data, arr[i].numel = k.encode(), t.data_ptr(), t.numel()
        h = C.c_void_p()
        N.check(L.vs_model_create(C.byref(c), arr, len(items), C.byref(h)), "vs_model_create")
        self._h, self._L = h, L
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.vs_model_destroy(self._h)
            self._h = None

    def _workspace(self, frames, H, W, step) -> torch.Tensor:
        need = int(self._L.vs_model_workspace_bytes(self._h, frames, H, W, step))
        if need < 0:
            raise N.NativeError("vs_model_workspace_bytes failed")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        return self._ws

    def embed(self, imgs: torch.Tensor, msgs: torch.Tensor, *, step: int = 1, video_mode: int = 0, lowres_attenuation: bool = False,
              antialias: bool = True, want_preds_w: bool = False):
        u8 = imgs.dtype == torch.uint8                 # RGB24 [F,H,W,3] in and out
        x = imgs.contiguous() if u8 else N.f32c(imgs)
        F_, H, W = (x.shape[0], x.shape[1], x.shape[2]) if u8 else (x.shape[0], x.shape[2], x.shape[3])
        m = msgs.to(device=x.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(x)
        pw = torch.empty(F_, self.cfg.out_ch, H, W, device=x.device) if want_preds_w else None
        ws = self._workspace(F_, H, W, step)
        N.check(self._L.vs_model_embed(self._h, N.ptr(x), N.ptr(m), m.shape[0], F_, H, W, step, video_mode, int(lowres_attenuation),
                                       int(antialias), int(u8), N.ptr(out), N.ptr(pw), N.ptr(ws), ws.numel(), N.stream()), "vs_model_embed")
        return (out, pw) if want_preds_w else out

    def detect(self, imgs: torch.Tensor, antialias: bool = True) -> torch.Tensor:
        u8 = imgs.dtype == torch.uint8
        x = imgs.contiguous() if u8 else N.f32c(imgs)
        F_, H, W = (x.shape[0], x.shape[1], x.shape[2]) if u8 else (x.shape[0], x.shape[2], x.shape[3])
        logits = torch.empty(F_, self.cfg.nbits + 1, device=x.device)
        ws = self._workspace(F_, H, W, 1)
        N.check(self._L.vs_model_detect(self._h, N.ptr(x), F_, H, W, int(antialias), int(u8), N.ptr(logits), N.ptr(ws), ws.numel(), N.stream()),
                "vs_model_detect")
        return logits
