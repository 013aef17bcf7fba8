"""Host-side driver of the HIP kernels: weight pre-packing and the launch sequence of the
U-Net embedder, the ConvNeXt-V2 extractor and the full-resolution shell.

Python here is plumbing only (pointer/shape bookkeeping, torch for device memory and streams);
every arithmetic stage of the hot path is a hand-written gfx950 kernel reached through the C-ABI
(native.py).  Weight packing (BN folding, NCHW -> [N][tap][C] re-layout) happens once per
state_dict on the device.

Reference stages replaced (file:line in /root/reference/videoseal):
  modules/unet.py:170-197  UNetMsg.forward            -> HipEngine.embedder_forward
  modules/convnext.py:146-156 + pixel_decoder.py:61-83 -> HipEngine.extractor_forward
  models/wam.py:161-197 / videoseal.py:303-344         -> HipEngine.resize_pre / jnd_lowres / embed_tail
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import native as N


def rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


TUNED_TILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_tiles_gfx950.json")


@dataclass
class Act:
    """NHWC fp32 activation: t is a flat/ND device tensor of B*H*W*ld floats."""
    t: torch.Tensor
    B: int
    H: int
    W: int
    C: int
    ld: int

    @property
    def rows(self) -> int:
        return self.B * self.H * self.W


@dataclass
class ConvW:
    wt: torch.Tensor          # [N][KH*KW*CinP]
    bias: Optional[torch.Tensor]
    N: int
    KH: int
    KW: int
    CinP: int
    split: Optional[torch.Tensor] = None   # [P][N][Ktot] 16-bit patterns (int16): P = 3 exact bf16 terms / P = 2 f16 terms of wt * w_mul
    blk: Optional[torch.Tensor] = None     # split weights in LDS-image order (producer/consumer kernels)
    arith: int = 3                         # arithmetic the planes were made for (vs_conv_desc_t::arith)
    w_mul: float = 1.0                     # power of two the weights were multiplied with before the split (arith 2)

    def _device_pack(self, arith: int) -> None:
        """planes + LDS image in ONE launch (csrc/pack.hip) -- bit-identical to split_f16x2 / split_bf16x3 + pack_blocked"""
        n, k = self.wt.shape
        P = 2 if arith == 2 else 3
        self.arith, self.w_mul = arith, (f16x2_scale(self.wt) if arith == 2 else 1.0)
        self.split = torch.empty(P, n, k, dtype=torch.int16, device=self.wt.device)
        self.blk = torch.empty((n + 31) // 32, k // 16, P, 64, 8, dtype=torch.int16, device=self.wt.device)
        N.check(N.lib().vs_split_block(N.ptr(self.wt), n, k, self.KH * self.KW, arith, float(self.w_mul), N.ptr(self.split), N.ptr(self.blk),
                                       N.stream()), "vs_split_block")

    def with_split(self, arith: int = 3) -> "ConvW":
        if self.split is None or self.arith != arith:
            if self.wt.is_cuda and self.wt.shape[1] % (16 * self.KH * self.KW) == 0:
                self._device_pack(arith)
                return self
            self.arith, self.blk = arith, None
            if arith == 2:
                self.split, self.w_mul = split_f16x2(self.wt)
            else:
                self.split, self.w_mul = split_bf16x3(self.wt), 1.0
        return self

    def with_blk(self, arith: int = 3) -> "ConvW":
        if self.blk is None or self.arith != arith:
            self.with_split(arith)
            if self.blk is None:
                self.blk = pack_blocked(self.split, self.KH * self.KW)
        return self


def pack_blocked(planes: torch.Tensor, ntaps: int = 1) -> torch.Tensor:
    """[P][N][Ktot] 16-bit patterns -> [ceil(N/32)][Ktot/16][P][64 slots][8]: every (32 rows x 16 k) block is 1 KiB in
    the order the kernel wants it in LDS; slot of (row r, k-half h) = 2r + (h ^ ((r>>3)&1)) (bank swizzle).
    Chunk order is (channel chunk, tap) -- the producer/consumer kernel walks all taps of a 16-channel chunk back to
    back so that the shifted re-reads of the same activation rows hit L1/L2 -- while Ktot is laid out (tap, channel)."""
    P, n, k = planes.shape
    G, nch = (n + 31) // 32, k // 16
    spt = nch // ntaps
    pad = torch.zeros(P, G * 32, k, dtype=planes.dtype, device=planes.device)
    pad[:, :n] = planes
    pad = pad.view(P, G * 32, ntaps, spt, 16).permute(0, 1, 3, 2, 4).reshape(P, G * 32, k)   # k order -> (chunk, tap, 16)
    x = pad.view(P, G, 32, nch, 2, 8).permute(1, 3, 0, 2, 4, 5).contiguous()       # [G][nch][P][r][h][8]
    r = torch.arange(32, device=planes.device)[:, None]
    h = torch.arange(2, device=planes.device)[None, :]
    slot = (2 * r + (h ^ ((r >> 3) & 1))).reshape(-1)                               # [(r,h)] -> slot
    out = torch.empty(G, nch, P, 64, 8, dtype=planes.dtype, device=planes.device)
    out[:, :, :, slot] = x.view(G, nch, P, 64, 8)
    return out.contiguous()


def split_bf16x3(w: torch.Tensor) -> torch.Tensor:
    """fp32 -> three bf16 terms by truncation, w == t1 + t2 + t3 exactly (8+8+8 mantissa bits).
    Returned as int16 bit patterns [3, *w.shape] (the kernel reads them as __bf16)."""
    w = w.float().contiguous()
    planes = []
    r = w
    for _ in range(3):
        hi = (r.view(torch.int32) & -65536).view(torch.float32)
        planes.append((hi.view(torch.int32) >> 16).to(torch.int16))
        r = r - hi
    return torch.stack(planes, 0).contiguous()


A_MUL = 16.0        # power of two the activations are multiplied with before the f16 split (vs_conv_desc_t::a_mul): |a| < 4094
A_MUL_GRN = 1.0     # the same for the GRN-scaled operand of pwconv2: |a| < 65504


def f16x2_scale(w: torch.Tensor) -> float:
    """the power of two of split_f16x2: max|w| * w_mul in [2^13, 2^14)"""
    amax = float(w.abs().max()) if w.numel() else 0.0
    kw = 14 - math.frexp(amax)[1] if amax > 0.0 and math.isfinite(amax) else 0
    return 2.0 ** max(-100, min(100, kw))


def split_f16x2(w: torch.Tensor) -> Tuple[torch.Tensor, float]:
    """fp32 -> two f16 terms of w * w_mul by round-to-nearest (hi = f16(w'), lo = f16(w' - hi)); w_mul is the power of two that puts
    max|w| into [2^13, 2^14) so that the low term of every weight >= 2^-16 max|w| is a normal f16 (conv_common.h, Arith<2>).
    Returned as int16 bit patterns [2, *w.shape] and w_mul."""
    w = w.float().contiguous()
    amax = float(w.abs().max()) if w.numel() else 0.0
    kw = 14 - math.frexp(amax)[1] if amax > 0.0 and math.isfinite(amax) else 0
    kw = max(-100, min(100, kw))
    ws = torch.ldexp(w, torch.tensor(kw, device=w.device))                     # exact
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    return torch.stack([hi.view(torch.int16), lo.view(torch.int16)], 0).contiguous(), 2.0 ** kw


def pack_conv(w: torch.Tensor, in_ld: int, scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, int]:
    """[N,Cin,KH,KW] -> [N][KH*KW*CinP] with k = (ky*KW+kx)*CinP + c; channels >= Cin are zero.
    `scale` (per output channel, e.g. folded BatchNorm) multiplies the rows."""
    n, cin, kh, kw = w.shape
    cinp = rup(max(in_ld, cin), 16)
    if w.is_cuda and scale is None:           # one launch (csrc/pack.hip) instead of zeros + permute + copy: a training step packs every layer
        wf = N.f32c(w)
        out = torch.empty(n, kh * kw * cinp, device=w.device, dtype=torch.float32)
        with torch.cuda.device(w.device):
            N.check(N.lib().vs_pack_conv(N.ptr(wf), n, cin, kh, kw, cinp, 0, N.ptr(out), N.stream()), "vs_pack_conv")
        return out, cinp
    out = torch.zeros(n, kh * kw, cinp, device=w.device, dtype=torch.float32)
    wk = w.float().permute(0, 2, 3, 1).reshape(n, kh * kw, cin)
    if scale is not None:
        wk = wk * scale.float()[:, None, None]
    out[:, :, :cin] = wk
    return out.reshape(n, kh * kw * cinp).contiguous(), cinp


def pack_conv_bwd(w: torch.Tensor, in_ld: int) -> Tuple[torch.Tensor, int]:
    """the backward-DATA weights of a conv [Co,Ci,KH,KW] (stride 1, 'same' padding) packed like pack_conv: rows = input channels, columns = output
    channels, taps flipped (dX = conv(dY, W')); for a 1x1 / Linear layer [N,K,1,1] the transposed GEMM matrix.  `in_ld`: row stride of dY."""
    co, ci, kh, kw = w.shape
    cinp = rup(max(in_ld, co), 16)
    wf = N.f32c(w)
    out = torch.empty(ci, kh * kw * cinp, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        N.check(N.lib().vs_pack_conv(N.ptr(wf), co, ci, kh, kw, cinp, 1, N.ptr(out), N.stream()), "vs_pack_conv")
    return out, cinp


def pack_patch_conv(w: torch.Tensor, pix_ld: int) -> Tuple[torch.Tensor, int]:
    """Non-/partly-overlapping k x k conv whose kx taps are contiguous in NHWC memory: the kw pixels of a
    row are read as one run of kw*pix_ld floats, so the conv becomes KH x 1 with Cin' = kw*pix_ld.
    [N,Cin,KH,KW] -> [N][KH*CinP'], k = ky*CinP' + kx*pix_ld + c."""
    n, cin, kh, kw = w.shape
    run = kw * pix_ld
    cinp = rup(run, 16)
    out = torch.zeros(n, kh, cinp, device=w.device, dtype=torch.float32)
    tmp = torch.zeros(n, kh, kw, pix_ld, device=w.device, dtype=torch.float32)
    tmp[..., :cin] = w.float().permute(0, 2, 3, 1)
    out[:, :, :run] = tmp.reshape(n, kh, run)
    return out.reshape(n, kh * cinp).contiguous(), cinp


def pack_cnx_block(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, beta: torch.Tensor, device) -> Tuple[torch.Tensor, float, float]:
    """Weight image of csrc/convnext_fused.hip for one ConvNeXt block: w1 [4C][C] (pwconv1), w2 [C][4C] (pwconv2), b1 / beta [4C].
    Per block of 32 h-channels: W1 rows as MFMA A fragments [k-step][plane][half][row 32][8], W2 as B fragments
    [n-block][k-step s][plane][half][n 32][8] with the reduction index enumerated as k(s, half, v) = 8 (2s + v // 4) + 4 half + v % 4 (the order in
    which the transposed pwconv1 leaves h in the accumulator registers), then b1[32], beta[32].  2 x f16 split (round to nearest even, on the CPU
    like the reference of pack.hip), per-matrix power-of-two scales.  Returns (uint8 image on `device`, w_mul1, w_mul2)."""
    w1, w2 = w1.float().cpu(), w2.float().cpu()
    C = w1.shape[1]
    H4 = w1.shape[0]
    nhb, ks1, nb = H4 // 32, C // 16, C // 32
    m1, m2 = f16x2_scale(w1), f16x2_scale(w2)

    def split(w, mul):
        ws = w * mul
        hi = ws.to(torch.float16)
        lo = (ws - hi.float()).to(torch.float16)
        return torch.stack([hi, lo], 0)                                   # [plane][...]
    s1 = split(w1, m1).view(2, nhb, 32, ks1, 2, 8)                        # (plane, hb, m, ks, half, v)
    img1 = s1.permute(1, 3, 0, 4, 2, 5).contiguous()                      # (hb, ks, plane, half, m, v)
    s2 = split(w2, m2).view(2, nb, 32, nhb, 2, 2, 2, 4)                   # (plane, nb, n, hb, s, vh, half, e): channel = 32 hb + 8 (2s + vh) + 4 half + e
    img2 = s2.permute(3, 1, 4, 0, 6, 2, 5, 7).contiguous()                # (hb, nb, s, plane, half, n, vh, e)
    aux = torch.zeros(nhb, 256, dtype=torch.float32)
    aux[:, :32] = b1.float().cpu().view(nhb, 32)
    aux[:, 32:64] = beta.float().cpu()[:H4].view(nhb, 32)
    img = torch.cat([img1.view(nhb, -1).view(torch.uint8), img2.view(nhb, -1).view(torch.uint8), aux.view(torch.uint8)], dim=1)
    return img.contiguous().to(device), m1, m2


def padvec(v: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, device=v.device, dtype=torch.float32)
    out[: v.numel()] = v.float().reshape(-1)
    return out


class HipEngine:
    """Packed weights + launch sequences for one architecture (ModelCfg) on one device."""
    arith = 2          # arithmetic of the split back-end (vs_conv_desc_t::arith); set per instance from VIDEOSEAL_CONV
    planes_chain = True
    planes_gemm = True
    # (class-level defaults: test harnesses build engines without a model, tests/test_gpu_kernels.py::Eng)
    _tr_pass = 0
    _calib = None
    thin_fused = True
    tail_variant = 0
    prof_extractor = False

    def __init__(self, cfg, sd: Dict[str, torch.Tensor], device: torch.device):
        self.cfg = cfg
        self.dev = device
        self.lib = N.lib()
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._ws_used: Dict[tuple, int] = {}      # 'tr.*' keys: the training pass that last touched them (begin_training_pass)
        self._tr_pass = 0
        self.kernel_timers = None        # list of (name, start_event, end_event, flops) when bench.py enables it
        self.prof_extractor = False      # time the extractor's stage-2 pwconv1 launches instead (detect-only workloads: no bottleneck conv)
        self.time_all_convs = False
        # arithmetic back-end of vs_conv_gemm: "f16x2" (2 x f16 split, 3 products), "bf16x3" (exact 3 x bf16 split, 6 products),
        # "f32" (v_mfma_f32_32x32x2_f32); "split" = the default split back-end
        conv_mode = os.environ.get("VIDEOSEAL_CONV", "auto")
        if conv_mode not in ("auto", "split", "f32", "bf16x3", "f16x2"):
            raise N.NativeError(f"VIDEOSEAL_CONV={conv_mode!r}: expected auto, f16x2, bf16x3 or f32")
        self.use_split = conv_mode != "f32"
        self.arith = 3 if conv_mode == "bf16x3" else 2
        # "auto" (default): 2 x f16 with an always-on range guard -- every network pass ends with a finite check of its (small) output on the
        # device (vs_check_finite).  The first pass with a new set of weights reads the flag synchronously: a network whose activations leave
        # the f16 range of the operand split (|a| * a_mul >= 65520: e.g. outlier channels of a trained GRN) is switched to the range-free exact
        # 3 x bf16 split and the pass is repeated, so the caller gets correct frames.  Later passes read the flag asynchronously (no sync in the
        # steady state); a data-dependent overflow then switches the network as well and is reported by the next API call.
        # "f16x2" forces the fast arithmetic without the guard (NaN frames on overflow; VIDEOSEAL_CHECK_FINITE=1 raises instead).
        self.auto_arith = conv_mode in ("auto", "split")
        self.arith_net = {"E": self.arith, "X": self.arith}
        self.verified = {"E": False, "X": False}
        # per-LAYER arithmetic of the ConvNeXt extractor: when the guard trips, one calibration pass on the exact split records the largest
        # operand of every dense layer (vs_absmax); only the layers whose operand leaves the f16 range (with a 4 x margin for other data) stay
        # on 3 x bf16, the rest of the network keeps the fast split -- one GRN outlier channel does not halve the whole extractor.
        # key (stage, block, 'pw1' | 'pw2') -> 3.  VIDEOSEAL_LAYER_ARITH=0: whole-network switch only.
        self.per_layer_arith = os.environ.get("VIDEOSEAL_LAYER_ARITH", "1") != "0"
        self.layer_arith: Dict[tuple, int] = {}
        self._calib: Optional[Dict[tuple, int]] = None        # during the calibration pass: key -> slot of _calib_bits
        self._calib_bits: Optional[torch.Tensor] = None
        self._nf_flag = torch.zeros(2, dtype=torch.int32, device=device)
        self._nf_host = torch.zeros(2, dtype=torch.int32).pin_memory() if device.type == "cuda" else None
        self._nf_event = None
        # Upsample groups as a low-resolution 9-tap GEMM + gather (a quarter of the MACs, no up-sampled concat); 0 = the literal
        # bilinear x2 -> reflect-pad conv3x3 -> LayerNorm sequence (kept for A/B checks)
        self.upconv_lowres = os.environ.get("VIDEOSEAL_UPCONV", "lowres") != "direct"
        self.upconv_fused = os.environ.get("VIDEOSEAL_UPCONV", "lowres") != "unfused"    # thin levels: GEMM + gather in one kernel
        self.msg_table_conv = os.environ.get("VIDEOSEAL_MSG_TABLE", "1") != "0"         # first bottleneck block: message channels as a table
        self.planes_chain = os.environ.get("VIDEOSEAL_PLANES", "1") != "0"              # bottleneck chain on pre-split operand planes
        self.msg0_planes = os.environ.get("VIDEOSEAL_MSG0_PLANES", "1") != "0"          # ... including its first block (round 5)
        # stage-2 pwconv2 on 128 x 96 tiles (tile code 26) with the whole K per workgroup instead of 2 K slices + the epilogue launch.  Neutral while
        # the wave-specialised GEMM's producers set its K loop's pace (profiles/r05a / r05c); with their lanes all working in both half steps the
        # kernel alone is 49.9 us against 56.3 + 13 (tools/bench_gemm.py ksweep2) and detect of 32 frames 3.725 -> 3.651 ms (median of five
        # alternating runs, profiles/r05o_detect_pw2_tile26.json).  VIDEOSEAL_PW2_NARROW=0: the K-slice form
        self.gemm_big = os.environ.get("VIDEOSEAL_GEMM_BIG", "0") == "1"                  # planes GEMMs of >= 3 rounds on 256 x 256 tiles, one wave per SIMD (round 6)
        self.stem_fused = os.environ.get("VIDEOSEAL_STEM_FUSED", "0") == "1"              # stem conv + LayerNorm in one VALU kernel (round 6; measured neutral: opt-in)
        self.down_patch = os.environ.get("VIDEOSEAL_DOWN_PATCH", "1") != "0"              # down-sampler LayerNorm -> patch matrix -> dense GEMM (round 6)
        self.grn_straddle = os.environ.get("VIDEOSEAL_GRN_STRADDLE", "1") != "0"         # GRN statistics from the planes GEMM's epilogue for HW % 32 != 0 (round 6)
        self.grn_fold = os.environ.get("VIDEOSEAL_GRN_FOLD", "1") != "0"                 # GRN finish inside the wave-specialised pwconv2 GEMM (round 6)
        self.pw2_narrow = os.environ.get("VIDEOSEAL_PW2_NARROW", "1") != "0"
        self.pw2_small_pc = os.environ.get("VIDEOSEAL_PW2_SMALL_PC", "1") != "0"        # stage-3 pwconv2 on the wave-specialised GEMM instead of planes (round 5)
        self.planes_gemm = os.environ.get("VIDEOSEAL_PLANES", "1") != "0"               # ConvNeXt 1x1 GEMMs on operand planes
        self.fused_blocks = os.environ.get("VIDEOSEAL_CNX_FUSED", "1") != "0"           # stage 0 / 1 blocks with h kept on chip (convnext_fused.hip)
        self.thin_fused = os.environ.get("VIDEOSEAL_THIN_FUSED", "1") != "0"              # 16-channel ResnetBlocks in one launch (resblock_thin.hip)
        self.planes_splitk = os.environ.get("VIDEOSEAL_PLANES_SPLITK", "1") != "0"      # planes chain with K slices for 4 - 8 key frames
        # VIDEOSEAL_CHECK_FINITE=1: synchronise after every network pass and raise if the output is not finite -- the 2 x f16 arithmetic
        # turns an activation beyond its f16 range (|a| * a_mul >= 65520) into inf / NaN instead of a silently wrong number
        self.check_finite = os.environ.get("VIDEOSEAL_CHECK_FINITE", "0") == "1"
        self.bn_sync = None       # callable(sums: float64 [2*ld + 1]) -> None, in-place sum over the ranks (dist.convert_sync_batchnorm)
        # per-shape tile selection: every candidate walks K in the same order, so the result is bit-identical whatever
        # tile wins -- only speed changes (measure, don't guess).  VIDEOSEAL_AUTOTUNE=0 keeps the static heuristic.
        self.autotune = os.environ.get("VIDEOSEAL_AUTOTUNE", "1") != "0"
        self.tail_variant = 0            # vs_tail_desc_t::variant (0 = default: row-streaming kernel where it applies)
        self.shell_timers: Optional[list] = None     # bench.py: (kernel, ev0, ev1, algorithmic bytes) of the HBM-bound shell kernels
        self._tile_cache: Dict[tuple, int] = {}
        # optional on-disk copy of the tile choices (VIDEOSEAL_TILE_CACHE=path.json): a profiled run then has no tuning launches
        # Packaged choices for the benchmark shapes (tools/tune_tiles.py, best of many interleaved rounds on an MI355X): a
        # quick in-process tuning is at the mercy of the box's clock wander, and ranks would otherwise tune independently.
        # Unknown signatures are still tuned on first use.
        self._tile_cache_path = os.environ.get("VIDEOSEAL_TILE_CACHE")
        self.tune_rounds = 2
        import json
        for path in (TUNED_TILES if self.autotune and os.environ.get("VIDEOSEAL_TUNED_TILES", "1") != "0" else None, self._tile_cache_path):
            if path and os.path.exists(path):
                with open(path) as f:
                    self._tile_cache.update({tuple(json.loads(k)): v for k, v in json.load(f).items()})
        # `sd` is the reference-format state_dict, or a callable returning the owner's live state_dict (model.py): tensors are
        # fetched when a weight group is (re)packed, so `invalidate()` after an optimiser step / load_state_dict re-reads them
        self._sd_src = sd if callable(sd) else (lambda: sd)
        self._sd_now = None
        self._g = self._fetch
        self.E = None          # packed embedder / extractor weights, built on first use (ChunkySeal: 1.0 G + 0.77 G parameters)
        self.Et = None         # embedder weights with BatchNorm NOT folded (batch-statistics forward under model.train())
        self.X = None
        self.ymat = self.taps43 = None
        self._pack_misc()

    def _fetch(self, k: str) -> torch.Tensor:
        if self._sd_now is None:
            self._sd_now = self._sd_src()
        return self._sd_now[k].detach().to(self.dev)

    def _pack_misc(self):
        g = self._g
        m = g("rgb2yuv.M").float().cpu()
        self.ymat = (C.c_float * 3)(*[float(v) for v in m[0]])
        if "attenuation.conv_lum.weight" in self._sd_now:
            taps = torch.cat([g("attenuation.conv_lum.weight")[0, 0].reshape(-1), g("attenuation.conv_x.weight")[0, 0].reshape(-1),
                              g("attenuation.conv_y.weight")[0, 0].reshape(-1)]).float().cpu()
            self.taps43 = (C.c_float * 43)(*[float(v) for v in taps])
        self._sd_now = None

    def invalidate(self, *groups: str):
        """drop packed weights: 'E' (folded embedder), 'Et' (train-mode embedder), 'X' (extractor), 'misc' (Y row, JND taps)"""
        self._sd_now = None
        for gname in (groups or ("E", "Et", "X", "misc")):
            if gname == "misc":
                self._pack_misc()
            else:
                setattr(self, gname, None)
                net = "X" if gname == "X" else "E"
                if self.auto_arith:                 # new weights: back to the fast arithmetic, to be verified by the next pass
                    self.arith_net[net], self.verified[net] = 2, False
                    self._nf_flag[0 if net == "E" else 1] = 0       # a flag raised by the OLD weights must not downgrade the new ones
                    self._nf_event = None
                if net == "X":
                    self.layer_arith.clear()

    # ------------------------------------------------------------------ packing
    def _bn_fold(self, g, p):
        s = g(p + ".weight").float() / torch.sqrt(g(p + ".running_var").float() + 1e-5)
        return s, g(p + ".bias").float() - g(p + ".running_mean").float() * s

    def _pack_resblock(self, g, p, cin, train=False):
        w0 = g(p + ".double_conv.0.weight")
        cout = w0.shape[0]
        wr, cpr = pack_conv(g(p + ".res_conv.weight"), rup(cin, 4))
        res = ConvW(wr, g(p + ".res_conv.bias").float().contiguous(), cout, 1, 1, cpr)
        if self.cfg.unet_norm == "rms":     # ChanRMSNorm + SiLU (legacy 0.0 card): nothing folds into the convs
            wt0, cp0 = pack_conv(w0, rup(cin, 4))
            wt1, cp1 = pack_conv(g(p + ".double_conv.3.weight"), rup(cout, 4))
            gm = [g(f"{p}.double_conv.{i}.gamma").float().reshape(-1).contiguous() for i in (1, 4)]
            return dict(c0=ConvW(wt0, None, cout, 3, 3, cp0), c1=ConvW(wt1, None, cout, 3, 3, cp1), res=res, cout=cout, rms=gm)
        if train:      # batch-statistics BatchNorm: raw convolutions + the live BN tensors (running stats are updated in place)
            wt0, cp0 = pack_conv(w0, rup(cin, 4))
            wt1, cp1 = pack_conv(g(p + ".double_conv.3.weight"), rup(cout, 4))
            bn = [dict(w=g(f"{p}.double_conv.{i}.weight").float().contiguous(), b=g(f"{p}.double_conv.{i}.bias").float().contiguous(),
                       rm=g(f"{p}.double_conv.{i}.running_mean"), rv=g(f"{p}.double_conv.{i}.running_var"),
                       nbt=g(f"{p}.double_conv.{i}.num_batches_tracked")) for i in (1, 4)]
            return dict(c0=ConvW(wt0, None, cout, 3, 3, cp0), c1=ConvW(wt1, None, cout, 3, 3, cp1), res=res, cout=cout, bn=bn)
        s0, b0 = self._bn_fold(g, p + ".double_conv.1")
        s1, b1 = self._bn_fold(g, p + ".double_conv.4")
        wt0, cp0 = pack_conv(w0, rup(cin, 4), s0)
        wt1, cp1 = pack_conv(g(p + ".double_conv.3.weight"), rup(cout, 4), s1)
        return dict(c0=ConvW(wt0, b0.contiguous(), cout, 3, 3, cp0), c1=ConvW(wt1, b1.contiguous(), cout, 3, 3, cp1),
                    res=res, cout=cout)

    def _pack_embedder(self, g, train=False):
        c = self.cfg
        u = "embedder.unet"
        zc = c.zc
        E = {}
        E["inc"] = self._pack_resblock(g, u + ".inc", c.in_ch, train)
        E["downs"] = []
        for i in range(len(zc) - 1):
            wd, cp = pack_conv(g(f"{u}.downs.{i}.down.weight"), rup(zc[i], 4))
            E["downs"].append(dict(down=ConvW(wd, g(f"{u}.downs.{i}.down.bias").float().contiguous(), zc[i + 1], 3, 3, cp),
                                   rb=self._pack_resblock(g, f"{u}.downs.{i}.conv", zc[i + 1], train)))
        E["bott"] = [self._pack_resblock(g, f"{u}.bottleneck.model.{j}", c.bott, train) for j in range(c.num_blocks)]
        zz = zc[:-1] + [c.bott]
        E["ups"] = []
        for k, i in enumerate(reversed(range(len(zz) - 1))):
            cin, cout = 2 * zz[i + 1], zz[i]
            wup = g(f"{u}.ups.{k}.up.upsample_block.2.weight")
            upd = dict(lnw=g(f"{u}.ups.{k}.up.upsample_block.3.weight").float().contiguous(),
                       lnb=g(f"{u}.ups.{k}.up.upsample_block.3.bias").float().contiguous(),
                       rb=self._pack_resblock(g, f"{u}.ups.{k}.conv", cout, train))
            if self.upconv_lowres and self.lib.vs_upconv_supported(cout):
                # per-tap products on the LOW-resolution map (vs_upconv_gather_ln): rows ordered (tap, channel), K = [x | skip]
                wz, cpz = pack_conv(wup.float().permute(2, 3, 0, 1).reshape(9 * cout, cin)[:, :, None, None], rup(cin, 4))
                upd["gemm"] = ConvW(wz, None, 9 * cout, 1, 1, cpz)
            else:
                wu, cp = pack_conv(wup, rup(cin, 4))
                upd["conv"] = ConvW(wu, None, cout, 3, 3, cp)
            E["ups"].append(upd)
        E["outc_w"] = g(u + ".outc.weight").float().reshape(c.out_ch, zc[0]).contiguous()
        E["outc_b"] = g(u + ".outc.bias").float().contiguous()
        E["table"] = g(u + ".msg_processor.msg_embeddings.weight").float().contiguous()
        for ch in zc + [c.bott]:
            if ch % 4:
                raise N.NativeError(f"U-Net channel count {ch} is not a multiple of 4 (unsupported by the HIP path)")
        if train:
            self.Et = E
        else:
            self.E = E
        self._sd_now = None

    @staticmethod
    def _xld(C_: int) -> int:
        """channel stride of the extractor's activations: wide layers are padded to whole 32-channel K pairs (zero lanes) so that the
        wave-specialised 1x1 GEMM applies to ChunkySeal's 362 / 724 / 1448 / 2896-channel stages as well"""
        return rup(C_, 32) if C_ >= 128 else rup(C_, 4)

    def _pack_extractor(self, g):
        c = self.cfg
        cn = "detector.convnext"
        d = c.dims
        X = {}
        ws, cps = pack_patch_conv(g(f"{cn}.downsample_layers.0.0.weight"), 4)
        X["stem"] = ConvW(ws, g(f"{cn}.downsample_layers.0.0.bias").float().contiguous(), d[0], 4, 1, cps)
        X["stem_ln"] = (g(f"{cn}.downsample_layers.0.1.weight").float().contiguous(), g(f"{cn}.downsample_layers.0.1.bias").float().contiguous())
        X["down"] = []
        for i in range(3):
            wd, cp = pack_patch_conv(g(f"{cn}.downsample_layers.{i+1}.1.weight"), self._xld(d[i]))
            X["down"].append(dict(lnw=g(f"{cn}.downsample_layers.{i+1}.0.weight").float().contiguous(),
                                  lnb=g(f"{cn}.downsample_layers.{i+1}.0.bias").float().contiguous(),
                                  conv=ConvW(wd, g(f"{cn}.downsample_layers.{i+1}.1.bias").float().contiguous(), d[i + 1], 2, 1, cp)))
            # round 6: the same rows [N][ky][kx * ld + c] read as ONE K run of 4 * ld = a 1x1 GEMM on the patch matrix vs_layernorm_patch2x2 writes
            X["down"][-1]["gemm"] = ConvW(wd, X["down"][-1]["conv"].bias, d[i + 1], 1, 1, 2 * cp)
        X["stages"] = []
        for st in range(4):
            blocks = []
            Cc = d[st]
            ld, ld4 = self._xld(Cc), self._xld(4 * Cc)
            for j in range(c.depths[st]):
                p = f"{cn}.stages.{st}.{j}"
                wdw = torch.zeros(49, ld, device=self.dev)
                wdw[:, :Cc] = g(p + ".dwconv.weight").float().reshape(Cc, 49).t()
                w1, cp1 = pack_conv(g(p + ".pwconv1.weight").float()[:, :, None, None], ld)
                w2, cp2 = pack_conv(g(p + ".pwconv2.weight").float()[:, :, None, None], ld4)
                fuse = None
                if st < 2 and Cc in (96, 192) and self.use_split and self.fused_blocks:
                    fuse = pack_cnx_block(g(p + ".pwconv1.weight"), g(p + ".pwconv1.bias"), g(p + ".pwconv2.weight"), g(p + ".grn.beta").reshape(-1), self.dev)
                blocks.append(dict(fuse=fuse, wdw=wdw.contiguous(), bdw=padvec(g(p + ".dwconv.bias"), ld), lnw=padvec(g(p + ".norm.weight"), ld),
                                   lnb=padvec(g(p + ".norm.bias"), ld),
                                   pw1=ConvW(w1, g(p + ".pwconv1.bias").float().contiguous(), 4 * Cc, 1, 1, cp1),
                                   gamma=g(p + ".grn.gamma").float().reshape(-1).contiguous(), beta=padvec(g(p + ".grn.beta"), rup(ld4, 16)),
                                   pw2=ConvW(w2, g(p + ".pwconv2.bias").float().contiguous(), Cc, 1, 1, cp2)))
            X["stages"].append(blocks)
        pd = "detector.pixel_decoder"
        wh, cph = pack_conv(g(pd + ".output_upscaling.0.upsample_block.2.weight"), self._xld(d[-1]))
        X["head_conv"] = ConvW(wh, None, d[-1], 3, 3, cph)
        X["head_ln"] = (g(pd + ".output_upscaling.0.upsample_block.3.weight").float().contiguous(), g(pd + ".output_upscaling.0.upsample_block.3.bias").float().contiguous())
        X["lin_w"] = g(pd + ".linear.weight").float().contiguous()
        X["lin_b"] = g(pd + ".linear.bias").float().contiguous()
        self.X = X
        self._sd_now = None

    # ------------------------------------------------------------------ workspace
    def buf(self, tag: str, numel: int, zero: bool = False) -> torch.Tensor:
        """named persistent workspace: stable addresses across calls of one shape (hipGraph-capturable)"""
        key = (tag, numel)
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(numel, device=self.dev, dtype=torch.float32) if zero else torch.empty(numel, device=self.dev, dtype=torch.float32)
            self._ws[key] = t
        if tag.startswith("tr."):
            self._ws_used[key] = self._tr_pass
        return t

    def begin_training_pass(self) -> None:
        """Called at the start of every training forward (EmbedTrainFn / DetectTrainFn / DetectorStep.step).  The training path's buffers
        ('tr.*': every saved activation of both networks) are keyed by size like all workspaces, but frame counts vary there (DropFrame /
        SpeedChange, a last partial batch, clips of different lengths) and a full activation set per distinct size would accumulate until
        the device is full: sizes that no pass of the last three touched are dropped.  The operands a pending backward needs are held by
        the references its saved dicts carry, so dropping the engine's handle never frees memory a graph still reads."""
        self._tr_pass += 1
        stale = [k for k, g in self._ws_used.items() if g < self._tr_pass - 3]
        for k in stale:
            self._ws.pop(k, None)
            del self._ws_used[k]

    def release_workspace(self):
        self._ws.clear()
        self._ws_used.clear()

    def new_act(self, tag: str, B: int, H: int, W: int, Cc: int, ld: Optional[int] = None) -> Act:
        ld = ld or rup(Cc, 4)
        return Act(self.buf(tag, B * H * W * ld), B, H, W, Cc, ld)

    # ------------------------------------------------------------------ kernel wrappers
    def conv(self, x: Act, w: ConvW, out: Act, *, stride=1, pad=0, pad_mode=N.PAD_ZERO, act=N.ACT_NONE, out_coff=0,
             n_store=None, res: Optional[Act] = None, in2: Optional[Act] = None, w2: Optional[ConvW] = None,
             a_scale=None, a_scale_ld=0, a_shift=None, geom=None, tile_hint=0, prof: Optional[str] = None,
             split_k: Optional[int] = None, sumsq: Optional[torch.Tensor] = None, cin: Optional[int] = None, flops: Optional[float] = None,
             grn_fold: Optional[tuple] = None, sumsq_hw: int = 0,
             arith: Optional[int] = None, in_pl: Optional[torch.Tensor] = None, in2_pl: Optional[torch.Tensor] = None,
             out_pl: Optional[torch.Tensor] = None, a_mul: Optional[float] = None):
        """cin: read only the first `cin` channels of every pixel (pixel stride stays x.ld)"""
        d = N.ConvDesc()
        if geom is None:
            sh = sw = stride
            ph = pw = pad
            H, W, sx = x.H, x.W, x.ld
            cin = x.ld if cin is None else cin
        else:           # patch conv: (W', sx, cin, sh, sw, ph, pw)
            W, sx, cin, sh, sw, ph, pw = geom
            H = x.H
        d.inp = N.ptr(x.t) if x.t is not None else None          # (None: the operand exists as planes only, see in_pl)
        d.in_sb, d.in_sy, d.in_sx = x.H * x.W * x.ld, x.W * x.ld, sx
        d.B, d.H, d.W, d.Cin = x.B, H, W, cin
        d.KH, d.KW, d.SH, d.SW, d.PH, d.PW, d.pad_mode = w.KH, w.KW, sh, sw, ph, pw, pad_mode
        d.Ho, d.Wo = out.H, out.W
        d.wt, d.CinP, d.N = N.ptr(w.wt), w.CinP, w.N
        d.a_scale, d.a_scale_ld, d.a_shift = N.ptr(a_scale), a_scale_ld, N.ptr(a_shift)
        d.bias, d.act = N.ptr(w.bias), act
        d.n_store = n_store if n_store is not None else (out.ld - out_coff if out_coff == 0 else w.N)
        if res is not None:
            d.res, d.res_ld = N.ptr(res.t), res.ld
        if in2 is not None:
            d.in2, d.in2_ld, d.Cin2, d.Cin2P = (N.ptr(in2.t) if in2.t is not None else None), in2.ld, in2.ld, w2.CinP
            d.wt2, d.bias2 = N.ptr(w2.wt), N.ptr(w2.bias)
        d.out, d.out_ld, d.out_coff, d.tile_hint = (N.ptr(out.t) if out.t is not None else None), out.ld, out_coff, tile_hint
        if self.use_split and not (tile_hint & N.CONV_FORCE_F32):
            ar = self.arith if arith is None else arith
            d.wt_split = N.ptr(w.with_blk(ar).split)
            d.wt_blk = N.ptr(w.blk)
            # activation range scale of the 2 x f16 split: 2^4 (|a| < 4094) for normalised tensors; the GRN-scaled pwconv2 operand
            # (h * (1 + gamma * Nx): outlier channels of a trained extractor can be large) keeps the whole f16 range (|a| < 65504)
            am = a_mul if a_mul is not None else (A_MUL_GRN if (a_scale is not None and not (tile_hint & N.CONV_PRE)) else A_MUL)
            d.arith, d.a_mul, d.acc_mul = ar, am, 1.0 / (am * w.w_mul)
            if in2 is not None:
                d.wt2_split = N.ptr(w2.with_blk(ar).split)
                d.wt2_blk = N.ptr(w2.blk)
                d.acc_mul2 = 1.0 / (am * w2.w_mul)
        if in_pl is not None:       # operands as pre-split planes (tile codes 22 / 23, conv3x3_pl.hip)
            d.in_pl, d.in2_pl, d.out_pl = N.ptr(in_pl), N.ptr(in2_pl), N.ptr(out_pl)
            if out.t is None:
                d.n_store = w.N
        if sumsq is not None:       # GRN partial sums of squares from the epilogue ([rows/32][N])
            d.sumsq_part = N.ptr(sumsq)
            d.sumsq_hw = int(sumsq_hw)       # > 0: frames of sumsq_hw rows (not a multiple of 32) -> the straddling [rows/32][2][N] form (tile codes 24 / 25)
            split_k = 1
        patch_pc = self._patch_pc_ok(d)
        if split_k is None:     # static, shape-only rule (never timing-based: a K split changes the summation order)
            split_k = 1
            if tile_hint == 0 and self._gemm_pc_ok(d):
                sk = self._split_k_rule(d)
                # the GRN rows of a K slice are staged in LDS (<= 3072 elements): the cap applies to the SLICE the rule picks (convnext_base stage 3,
                # K = 4096, runs as two slices of 2048); a slice beyond it leaves the launch unsplit for vs_conv_gemm's generic dispatch.
                # model_api.hip mirrors this rule line for line, so the Python engine and the C-ABI host sum K in the same order
                if not d.a_scale or d.CinP // sk <= 3072:
                    split_k = sk
            elif tile_hint == 0 and patch_pc:
                split_k = self._split_k_rule_patch(d)
        if split_k > 1:
            ws_ld = rup(w.N, 4)
            slices = split_k + (1 if ((patch_pc or in_pl is not None) and in2 is not None) else 0)     # + the slice of the 1x1 second phase
            d.splitk_ws, d.splitk_ld, d.split_k = N.ptr(self.buf(f"splitk.ws@{N.stream()}", slices * out.rows * ws_ld)), ws_ld, split_k
            if tile_hint == 0 and not self.autotune:
                d.tile_hint = self._static_split_tile(d)
        if tile_hint == 0 and self.autotune:
            d.tile_hint = self._pick_tile(d, w, out)
        if grn_fold is not None:
            # grn_fold = (partials [B][HW/32][K] from pwconv1's epilogue, gamma, B, HW): GRN's finish (||h|| -> scale) either INSIDE this GEMM
            # (round 6, csrc/gemm1x1_pc.hip: the launch is known to run on the wave-specialised kernel, tile codes 17 / 18 / 26, and a frame has
            # <= 16 partial rows) or as the separate launch it always was.  Same scale values bit for bit either way
            part, gamma, B_, HW_, C4 = grn_fold
            pc_tile = (d.tile_hint & 0xf) + (16 if d.tile_hint & N.CONV_TILE_HI else 0)
            if self.grn_fold and pc_tile in (17, 18, 26) and HW_ % 32 == 0 and HW_ // 32 <= 16 and w.CinP == a_scale_ld == C4:
                d.grn_part, d.grn_gamma, d.grn_nchunk = N.ptr(part), N.ptr(gamma), HW_ // 32
            else:
                N.check(self.lib.vs_grn_scale_from_partials(N.ptr(part), B_, HW_, C4, N.ptr(gamma), N.ptr(a_scale), a_scale_ld, N.stream()),
                        "vs_grn_scale_from_partials")
        if self.kernel_timers is not None and prof is None and self.time_all_convs:
            prof = f"conv{w.KH}x{w.KW} {x.C}->{w.N} @{out.H}x{out.W}" + ("+1x1" if in2 is not None else "")
        timed = prof is not None and self.kernel_timers is not None and not torch.cuda.is_current_stream_capturing()
        if timed:   # HIP events on the launch stream, used by bench.py for the per-kernel roofline
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        N.check(self.lib.vs_conv_gemm(C.byref(d), N.stream()), "vs_conv_gemm")
        if timed:
            ev1.record()
            k_total = w.KH * w.KW * w.CinP + (w2.CinP if w2 is not None else 0)
            self.kernel_timers.append((prof, ev0, ev1, 2.0 * out.rows * w.N * k_total if flops is None else flops))
        return out

    @staticmethod
    def _gemm_pc_ok(d: "N.ConvDesc") -> bool:
        """preconditions of the wave-specialised 1x1 GEMM (tile codes 17 / 18), mirrored from vs_conv_gemm"""
        return (bool(d.wt_split) and bool(d.wt_blk) and d.KH == 1 and d.KW == 1 and d.SH == 1 and d.SW == 1 and d.PH == 0 and d.PW == 0
                and not d.in2 and d.Ho == d.H and d.Wo == d.W and d.Cin % 32 == 0 and d.CinP == d.Cin
                and d.in_sy == d.W * d.in_sx and d.in_sb == d.H * d.in_sy and (not d.a_scale or d.H * d.W >= 128 or d.H * d.W == 64))

    @staticmethod
    def _patch_pc_ok(d: "N.ConvDesc") -> bool:
        """preconditions of the wave-specialised 3x3 patch kernel (tile codes 15 / 16), mirrored from vs_conv_gemm"""
        return (bool(d.wt_split) and bool(d.wt_blk) and d.KH == 3 and d.KW == 3 and d.SH == 1 and d.SW == 1 and d.PH == 1 and d.PW == 1
                and d.Ho == d.H and d.Wo == d.W and not d.a_scale and d.W % 16 == 0 and d.H % 8 == 0 and (not d.in2 or bool(d.wt2_blk)))

    @staticmethod
    def _split_k_rule_patch(d: "N.ConvDesc") -> int:
        """K slices (whole 16-channel chunks) for wide 3x3 layers with too few 128-pixel tiles to fill 256 CUs,
        e.g. the 4 key frames of a 16-frame streaming chunk: 32 tiles x 2 = 64 workgroups."""
        if d.N < 128 or d.CinP < 128:
            return 1
        blocks = d.B * (d.H // 8) * (d.W // 16) * ((d.N + 191) // 192 if d.N % 192 == 0 else (d.N + 127) // 128)
        spt = d.CinP // 16
        sk = 1
        for cand in (2, 3, 4, 6, 8):
            if blocks * sk >= 192:
                break
            if spt % cand == 0 and spt // cand >= 2:
                sk = cand
        return sk

    @staticmethod
    def _static_split_tile(d: "N.ConvDesc") -> int:
        if d.KH == 3:
            return N.CONV_TILE_HI | 0 if d.N % 192 == 0 else 15
        return N.CONV_TILE_HI | (2 if d.N % 192 == 0 else 1)

    @staticmethod
    def _split_k_rule(d: "N.ConvDesc") -> int:
        """K slices for small-M GEMMs: double while 128 x 128 tiles cannot give every CU a workgroup and a slice keeps >= 4 K pairs"""
        rows = d.B * d.H * d.W
        blocks = ((rows + 127) // 128) * ((d.N + 127) // 128)
        pairs = d.CinP // 32
        sk = 1          # (aiming at 128 / 512 / 1024 workgroups instead of 256 measured slower: detect of 32 frames 4.17 / 3.95 / 4.24 against 3.87 ms, round 4)
        while blocks * sk < 256 and pairs % (sk * 2) == 0 and pairs // (sk * 2) >= 4:
            sk *= 2
        return sk

    def _pick_tile(self, d: "N.ConvDesc", w: ConvW, out: Act) -> int:
        """time the 4-wave tile shapes once per conv signature (on a scratch output) and remember the fastest."""
        key = (d.B, d.H, d.W, d.Cin, d.KH, d.KW, d.SH, d.SW, d.pad_mode, d.Ho, d.Wo, d.N, d.CinP, bool(d.in2), d.Cin2P,
               bool(d.a_scale), bool(d.res), d.act, bool(d.wt_split), d.split_k, bool(d.sumsq_part))
        if d.wt_split and d.arith == 2:
            key += (2,)          # the 2 x f16 arithmetic has its own timings (half the MFMA work per tile)
        best = self._tile_cache.get(key)
        if best is not None:
            return best
        if torch.cuda.is_current_stream_capturing():
            # never time inside a hipGraph capture: static heuristic (same numerics)
            return self._static_split_tile(d) if d.split_k > 1 else 0
        patch = (bool(d.wt_split) and d.KH == 3 and d.KW == 3 and d.SH == 1 and d.SW == 1 and d.PH == 1 and d.PW == 1 and
                 d.Ho == d.H and d.Wo == d.W and not d.a_scale and d.W % 16 == 0 and d.H % 8 == 0)
        small = (bool(d.wt_split) and d.KH == 3 and d.KW == 3 and d.SH == 1 and d.SW == 1 and d.PH == 1 and d.PW == 1 and d.Ho == d.H
                 and d.Wo == d.W and not d.a_scale and d.CinP == 16 and d.N <= 32 and d.n_store <= 32 and (not d.in2 or d.Cin2P == 16)
                 and d.split_k <= 1 and not d.sumsq_part)
        if patch and d.split_k > 1:
            cands = [15] + ([N.CONV_TILE_HI] if d.N > 128 else [])
        elif patch and small:     # + the persistent thin-layer kernel (tile 20); same K order -> bit-identical
            cands = [10, N.CONV_TILE_HI | 4]
        elif patch:     # the patch kernel walks K as (chunk, tap): candidates stay inside one K order (bit-identical results)
            # 15 / TILE_HI|0 (=16): wave-specialised variants, 128 and 192 output channels per workgroup
            widths = {10: 32, 11: 64, 12: 128, 15: 128, N.CONV_TILE_HI: 192}
            cands = [t for t in widths if widths[t] < 2 * d.N + 64 or t == 10]
            if 32 < d.N <= 64 and d.H % 16 == 0 and d.CinP >= 64:
                cands.append(N.CONV_TILE_HI | 5)      # tile 21: 256 pixels x 64 channels (wave-specialised)
        elif d.split_k > 1:      # K-split plan fixed by the shape rule: only the kernels that implement it
            cands = [N.CONV_TILE_HI | 1] + ([N.CONV_TILE_HI | 2] if d.N > 128 else [])
        else:
            cands = [t for t in (1, 2, 3, 4, 5, 13, 14) if self._tile_ok(t, d.N)]
            if self._gemm_pc_ok(d) and d.N >= 96:    # same K order as the generic kernel: bit-identical candidates
                cands += [N.CONV_TILE_HI | 1] + ([N.CONV_TILE_HI | 2] if d.N > 128 else [])
        real_out, real_coff, real_ld = d.out, d.out_coff, d.out_ld
        scratch = self.buf("autotune.out", out.rows * rup(d.n_store, 4))
        d.out, d.out_coff, d.out_ld = N.ptr(scratch), 0, rup(d.n_store, 4)
        times = {t: float("inf") for t in cands}
        for rnd in range(self.tune_rounds):          # candidates interleaved, best round of each: the box's clocks wander by ~10 %
            for t in cands:
                d.tile_hint = t
                if rnd == 0:
                    N.check(self.lib.vs_conv_gemm(C.byref(d), N.stream()), "vs_conv_gemm(autotune)")       # warm
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    N.check(self.lib.vs_conv_gemm(C.byref(d), N.stream()), "vs_conv_gemm(autotune)")
                e1.record()
                e1.synchronize()
                times[t] = min(times[t], e0.elapsed_time(e1))
        best = min(times, key=times.get)
        d.out, d.out_coff, d.out_ld = real_out, real_coff, real_ld
        self._tile_cache[key] = best
        if self._tile_cache_path:
            import json
            with open(self._tile_cache_path, "w") as f:
                json.dump({json.dumps(list(k)): v for k, v in self._tile_cache.items()}, f)
        return best

    @staticmethod
    def _tile_ok(tile: int, n: int) -> bool:
        bn = {1: 128, 2: 64, 3: 32, 4: 192, 5: 96, 13: 64, 14: 128}[tile]
        return bn < 2 * n + 64 or tile == 3     # skip tiles that would be mostly padding

    def to_planes(self, x: Act, tag: str) -> torch.Tensor:
        """fp32 NHWC activation -> the operand planes of the all-DMA 3x3 kernel: int16 [2][C/16][rows][16] (f16 hi / lo of x * A_MUL)"""
        if x.C % 16:
            raise N.NativeError("planes need a channel count that is a multiple of 16")
        pl = self.buf(tag, x.rows * x.C).view(torch.int16)[: 2 * x.rows * x.C]
        N.check(self.lib.vs_to_planes(N.ptr(x.t), x.rows, x.C, x.ld, A_MUL, N.ptr(pl), N.stream()), "vs_to_planes")
        return pl

    def layernorm(self, x: Act, w, b, out: Act, act=N.ACT_NONE, eps=1e-6):
        N.check(self.lib.vs_layernorm_act(N.ptr(x.t), x.rows, x.C, x.ld, N.ptr(w), N.ptr(b), eps, act, N.ptr(out.t), out.ld,
                                          N.stream()), "vs_layernorm_act")
        return out

    def _bn_batch(self, raw: Act, bn: dict, act: int, out: Act, add: Optional[Act] = None):
        """nn.BatchNorm2d training branch on an NHWC tensor + activation (+ the res_conv branch): batch statistics, running
        statistics updated in place (momentum 0.1, unbiased variance), num_batches_tracked += 1.  With `self.bn_sync` set
        (videoseal_amd.dist.convert_sync_batchnorm) the statistics are those of the global batch: nn.SyncBatchNorm, train.py:438-440."""
        L, st = self.lib, N.stream()
        part = self.buf("bn.part", 2 * int(L.vs_bn_partial_doubles(raw.rows, raw.ld)))       # doubles = 2 floats each
        ss = self.buf("bn.ss", 2 * raw.ld)
        scale, shift = ss[: raw.ld], ss[raw.ld:]
        if self.bn_sync is not None:
            sums = self.buf("bn.sums", 2 * (2 * raw.ld + 2)).view(torch.float64)[: 2 * raw.ld + 1]     # [sum x | sum x^2 | rows]
            N.check(L.vs_bn_partial_sums(N.ptr(raw.t), raw.rows, raw.C, raw.ld, N.ptr(part), N.ptr(sums), st), "vs_bn_partial_sums")
            self.bn_sync(sums)                       # ONE all-reduce of 2*ld + 1 doubles per BatchNorm layer (in place, same stream)
            N.check(L.vs_bn_finish_sums(N.ptr(sums), raw.C, raw.ld, N.ptr(bn["w"]), N.ptr(bn["b"]), 1e-5, 0.1, N.ptr(bn["rm"]),
                                        N.ptr(bn["rv"]), N.ptr(scale), N.ptr(shift), st), "vs_bn_finish_sums")
        else:
            N.check(L.vs_bn_batch_stats(N.ptr(raw.t), raw.rows, raw.C, raw.ld, N.ptr(bn["w"]), N.ptr(bn["b"]), 1e-5, 0.1, N.ptr(bn["rm"]),
                                        N.ptr(bn["rv"]), N.ptr(part), N.ptr(scale), N.ptr(shift), st), "vs_bn_batch_stats")
        bn["nbt"].add_(1)
        N.check(L.vs_scale_shift_act(N.ptr(raw.t), raw.rows, raw.C, raw.ld, N.ptr(scale), N.ptr(shift), act,
                                     N.ptr(add.t) if add is not None else None, add.ld if add is not None else 0, N.ptr(out.t), out.ld, st),
                "vs_scale_shift_act")
        return out

    def resblock_train(self, x: Act, p, tag: str, out: Optional[Act] = None) -> Act:
        """unet.py:17-39 under model.train(): conv -> BN(batch stats) -> ReLU twice, + res_conv(x)."""
        cout = p["cout"]
        raw = self.new_act(tag + ".raw", x.B, x.H, x.W, cout)
        t = self.new_act(tag + ".t", x.B, x.H, x.W, cout)
        self.conv(x, p["c0"], raw, pad=1)
        self._bn_batch(raw, p["bn"][0], N.ACT_RELU, t)
        self.conv(t, p["c1"], raw, pad=1)
        rs = self.new_act(tag + ".res", x.B, x.H, x.W, cout)
        self.conv(x, p["res"], rs)
        if out is None:
            out = self.new_act(tag + ".o", x.B, x.H, x.W, cout)
        return self._bn_batch(raw, p["bn"][1], N.ACT_RELU, out, add=rs)

    def resblock_rms(self, x: Act, p, tag: str, out: Optional[Act] = None) -> Act:
        """unet.py:17-39 with ChanRMSNorm + SiLU (common.py:118-119, 172-179): silu(rms(conv(silu(rms(conv(x)))))) + res_conv(x)."""
        cout, L, st = p["cout"], self.lib, N.stream()
        raw = self.new_act(tag + ".raw", x.B, x.H, x.W, cout)
        t = self.new_act(tag + ".t", x.B, x.H, x.W, cout)
        self.conv(x, p["c0"], raw, pad=1)
        N.check(L.vs_rmsnorm_act(N.ptr(raw.t), raw.rows, cout, raw.ld, N.ptr(p["rms"][0]), N.ACT_SILU, None, 0, N.ptr(t.t), t.ld, st), "vs_rmsnorm_act")
        self.conv(t, p["c1"], raw, pad=1, prof=("bott.conv3x3" if tag.startswith("bott") else None))     # (the legacy card's dominant launch, for bench.py's roofline)
        rs = self.new_act(tag + ".res", x.B, x.H, x.W, cout)
        self.conv(x, p["res"], rs)
        if out is None:
            out = self.new_act(tag + ".o", x.B, x.H, x.W, cout)
        if out.ld != rup(cout, 4):
            raise N.NativeError("resblock_rms writes whole rows")
        N.check(L.vs_rmsnorm_act(N.ptr(raw.t), raw.rows, cout, raw.ld, N.ptr(p["rms"][1]), N.ACT_SILU, N.ptr(rs.t), rs.ld, N.ptr(out.t), out.ld, st),
                "vs_rmsnorm_act")
        return out

    def resblock_msg0(self, h3: Act, p, tag: str, lat: torch.Tensor, Bm: int, nlat: int, planes_out: bool = False):
        """planes_out: the caller continues with bottleneck_planes -> returns (activation, its operand planes or None).
        First bottleneck block (unet.py:183-185) on h3 = [latent (nlat channels) | message (spatially constant)]: the 3x3 conv over
        the message channels is a per-frame table of nine border classes (vs_msg_pre), so the conv's K loop covers the latent channels
        only (9*128 instead of 9*384 for VideoSeal 1.0) and the table is added before bias + activation (VS_CONV_PRE)."""
        cout, c0 = p["cout"], p["c0"]
        hidden = h3.C - nlat
        if "c0_lat" not in p:          # sliced once from the BN-folded packed weight [N][9][CinP]
            w3 = c0.wt.view(cout, 9, c0.CinP)
            p["c0_lat"] = ConvW(w3[:, :, :nlat].reshape(cout, 9 * nlat).contiguous(), c0.bias, cout, 3, 3, nlat)
            wm = w3[:, :, nlat:nlat + hidden].permute(1, 0, 2).reshape(9 * cout, hidden).contiguous()      # rows (tap, n)
            p["c0_msg"] = ConvW(wm, None, 9 * cout, 1, 1, hidden)
        P = Act(self.buf(tag + ".P", Bm * 9 * cout), Bm, 1, 1, 9 * cout, 9 * cout)
        self.conv(Act(lat, Bm, 1, 1, hidden, hidden), p["c0_msg"], P)
        pre = self.buf(tag + ".pre", Bm * 9 * cout)          # border-class table [Bm][9][N]
        N.check(self.lib.vs_msg_pre(N.ptr(P.t), Bm, cout, N.ptr(pre), N.stream()), "vs_msg_pre")
        tile = (N.CONV_TILE_HI | 0) if cout % 192 == 0 else 15
        pre_ld = 0 if Bm == 1 else 9 * cout
        prof0 = "bott.conv3x3.lat" if self.time_all_convs else None
        ghost = Act(None, h3.B, h3.H, h3.W, cout, cout)
        sk = self._planes_split(ghost) if (planes_out and self.msg0_planes and cout == h3.C and cout % 192 == 0 and nlat % 16 == 0 and
                                           h3.H % 16 == 0 and h3.W % 16 == 0) else 0
        if sk >= 1:
            # round 5: the block on the all-DMA planes kernel like the rest of the chain (it ran on the wave-specialised kernel: 155 + 359 us
            # of an image step against 228 us for the same c1 launch on planes).  The latent channels and the whole of h3 are split once
            # (two conversion passes, 8 + 20 us), c0 reads the latent planes and adds the border-class table in its epilogue (VS_CONV_PRE),
            # c1 reads t and h3 (its 1x1 second K phase) as planes and writes the chain's first operand planes: no fp32 t / out, and
            # bottleneck_planes no longer converts its input.  Same products and K order as the kernels it replaces -> same bits at sk = 1.
            # With few key frames (sk > 1: K slices) c0 keeps the wave-specialised kernel -- the table epilogue has no K-slice form.
            ptile = N.CONV_TILE_HI | 6
            h3pl = self.to_planes(h3, tag + ".h3pl")
            tpl = self.buf("bott.plt", h3.rows * cout).view(torch.int16)
            xpl = self.buf("bott.pl0", h3.rows * cout).view(torch.int16)
            if sk == 1:
                latpl = self.buf(tag + ".latpl", h3.rows * nlat).view(torch.int16)[: 2 * h3.rows * nlat]
                N.check(self.lib.vs_to_planes(N.ptr(h3.t), h3.rows, nlat, h3.ld, A_MUL, N.ptr(latpl), N.stream()), "vs_to_planes")
                self.conv(Act(None, h3.B, h3.H, h3.W, nlat, nlat), p["c0_lat"], ghost, pad=1, act=N.ACT_RELU, a_scale=pre, a_scale_ld=pre_ld,
                          tile_hint=ptile | N.CONV_PRE, in_pl=latpl, out_pl=tpl, prof=prof0)
            else:
                t = self.new_act(tag + ".t", h3.B, h3.H, h3.W, cout)
                self.conv(h3, p["c0_lat"], t, pad=1, act=N.ACT_RELU, cin=nlat, a_scale=pre, a_scale_ld=pre_ld, tile_hint=tile | N.CONV_PRE, prof=prof0)
                N.check(self.lib.vs_to_planes(N.ptr(t.t), t.rows, cout, t.ld, A_MUL, N.ptr(tpl), N.stream()), "vs_to_planes")
            self.conv(ghost, p["c1"], ghost, pad=1, act=N.ACT_RELU, in2=Act(None, h3.B, h3.H, h3.W, h3.C, h3.C), w2=p["res"], tile_hint=ptile,
                      in_pl=tpl, in2_pl=h3pl, out_pl=xpl, **(dict(split_k=sk) if sk > 1 else {}))
            self.planes_chain_ran = True
            return ghost, xpl
        t = self.new_act(tag + ".t", h3.B, h3.H, h3.W, cout)
        self.conv(h3, p["c0_lat"], t, pad=1, act=N.ACT_RELU, cin=nlat, a_scale=pre, a_scale_ld=pre_ld, tile_hint=tile | N.CONV_PRE, prof=prof0)
        out = self.new_act(tag + ".o", h3.B, h3.H, h3.W, cout)
        self.conv(t, p["c1"], out, pad=1, act=N.ACT_RELU, in2=h3, w2=p["res"])
        return (out, None) if planes_out else out

    def _msg0_ok(self, h3: Act, p, nlat: int) -> bool:
        hidden = h3.C - nlat
        return (self.use_split and self.msg_table_conv and "bn" not in p and "rms" not in p and nlat % 16 == 0 and hidden % 32 == 0 and
                h3.ld == h3.C and p["c0"].CinP == h3.C and h3.W % 16 == 0 and h3.H % 8 == 0 and p["cout"] >= 128 and
                h3.B * (h3.H // 8) * (h3.W // 16) >= 128)

    def resblock(self, x: Act, p, tag: str, out: Optional[Act] = None, out_coff=0) -> Act:
        """unet.py:38-39  relu(bn(conv(relu(bn(conv(x)))))) + res_conv(x); the 1x1 rides in the 2nd conv's K loop."""
        if "rms" in p:
            assert out_coff == 0
            return self.resblock_rms(x, p, tag, out)
        if "bn" in p:
            assert out_coff == 0
            return self.resblock_train(x, p, tag, out)
        cout = p["cout"]
        if out_coff == 0 and self._thin_ok(x, p, out):
            return self.resblock_thin(x, p, tag, out)
        t = self.new_act(tag + ".t", x.B, x.H, x.W, cout)
        self.conv(x, p["c0"], t, pad=1, act=N.ACT_RELU, prof=("bott.conv3x3" if tag.startswith("bott") else None))
        if out is None:
            out = self.new_act(tag + ".o", x.B, x.H, x.W, cout)
        self.conv(t, p["c1"], out, pad=1, act=N.ACT_RELU, in2=x, w2=p["res"], out_coff=out_coff,
                  n_store=(cout if out.ld != rup(cout, 4) else None))
        return out

    def _thin_ok(self, x: Act, p, out: Optional[Act]) -> bool:
        """the whole ResnetBlock in one launch with t on chip (csrc/resblock_thin.hip): 16 mid / output channels from <= 16 input channels, or 32
        from a 32-channel map (2 x f16 arithmetic only: its weights live in LDS)"""
        c = p["cout"]
        if not (self.thin_fused and self.use_split and "bn" not in p and "rms" not in p and c in (16, 32)):
            return False
        k = 16 if c == 16 else 32
        return (x.ld <= k and p["c0"].CinP == k and p["c1"].CinP == k and p["res"].CinP == k and (out is None or out.ld == c) and
                (c == 16 or self.arith == 2) and bool(self.lib.vs_resblock_thin_supported(x.ld, c, c)))

    def resblock_thin(self, x: Act, p, tag: str, out: Optional[Act] = None) -> Act:
        ar, c = self.arith, p["cout"]
        c0, c1, cr = p["c0"].with_split(ar), p["c1"].with_split(ar), p["res"].with_split(ar)
        if out is None:
            out = self.new_act(tag + ".o", x.B, x.H, x.W, c)
        d = N.ResblockThinDesc()
        d.x, d.x_ld, d.B, d.H, d.W, d.Cin = N.ptr(x.t), x.ld, x.B, x.H, x.W, x.ld
        d.w0_split, d.w1_split, d.wr_split = N.ptr(c0.split), N.ptr(c1.split), N.ptr(cr.split)
        d.b0, d.b1, d.br = N.ptr(c0.bias), N.ptr(c1.bias), N.ptr(cr.bias)
        d.arith, d.Cout, d.a_mul = ar, c, A_MUL
        d.acc_mul0, d.acc_mul1, d.acc_mulr = 1.0 / (A_MUL * c0.w_mul), 1.0 / (A_MUL * c1.w_mul), 1.0 / (A_MUL * cr.w_mul)
        d.out, d.out_ld = N.ptr(out.t), out.ld
        timed = self.kernel_timers is not None and self.time_all_convs and not torch.cuda.is_current_stream_capturing()
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        N.check(self.lib.vs_resblock_thin(C.byref(d), N.stream()), "vs_resblock_thin")
        if timed:
            ev1.record()
            self.kernel_timers.append((f"resblock_thin {x.C}->{c}->{c} @{x.H}x{x.W}", ev0, ev1, 2.0 * x.rows * c * (2 * 9 * c + c)))
        return out

    _WHY = (f"with the 2 x f16 arithmetic an activation outside the f16 range of the operand split (|a| >= {65520.0 / A_MUL:.0f}; "
            f"{65520.0 / A_MUL_GRN:.0f} for the GRN-scaled pwconv2 input) becomes inf; VIDEOSEAL_CONV=bf16x3 selects the range-free exact split")

    def _guard(self, net: str, t: torch.Tensor) -> bool:
        """finite check of a network's output (see __init__).  False = the pass must be repeated (the network was just switched to 3 x bf16)."""
        idx = 0 if net == "E" else 1
        guarded = self.use_split and self.arith_net[net] == 2 and self.auto_arith
        if not guarded:
            if self.check_finite and not bool(torch.isfinite(t).all()):
                raise N.NativeError(f"{'embedder' if net == 'E' else 'extractor'}: non-finite output -- " + self._WHY)
            return True
        N.check(self.lib.vs_check_finite(N.ptr(t), t.numel(), self._nf_flag.data_ptr() + 4 * idx, N.stream()), "vs_check_finite")
        if torch.cuda.is_current_stream_capturing():
            return True
        if not self.verified[net]:
            if int(self._nf_flag[idx].item()):          # one synchronisation per set of weights
                self._nf_flag[idx] = 0
                self.arith_net[net] = 3
                import warnings
                warnings.warn(f"{'embedder' if net == 'E' else 'extractor'}: activations leave the f16 range of the 2 x f16 operand split with these "
                              f"weights; this network now runs on the exact, range-free 3 x bf16 split (twice the matrix work)")
                return False
            self.verified[net] = True
            return True
        self.note_guard()
        return True

    def note_guard(self) -> None:
        """steady state (verified weights) and hipGraph replays: copy the guard flags to pinned host memory behind the pass, no synchronisation"""
        if not (self.use_split and self.auto_arith) or self._nf_host is None:
            return
        self._nf_host.copy_(self._nf_flag, non_blocking=True)
        self._nf_event = torch.cuda.Event()
        self._nf_event.record()

    def _take_guard(self) -> List[str]:
        """networks whose (already delivered) guard flags say 'non-finite': switched to 3 x bf16 here, flags cleared"""
        self._nf_event = None
        bad = [n for i, n in enumerate(("E", "X")) if int(self._nf_host[i]) and self.arith_net[n] == 2]
        if bad:
            for n in bad:
                self.arith_net[n] = 3
            self._nf_flag.zero_()
            self._nf_host.zero_()
        return bad

    def poll_nonfinite(self) -> None:
        """API entry: has an earlier (already verified) pass tripped the range guard?  Then its frames were non-finite: switch the network to
        3 x bf16 for everything that follows and tell the caller (no silent NaN frames)."""
        ev = self._nf_event
        if ev is None or not ev.query():
            return
        bad = self._take_guard()
        if bad:
            raise N.NativeError("an earlier call produced non-finite values in the " + " and ".join("embedder" if n == "E" else "extractor" for n in bad) +
                                " (data-dependent overflow): " + self._WHY + ".  The engine has switched to it for this model: repeat the call.")

    def guard_tripped_after_sync(self) -> bool:
        """For entry points that synchronise anyway before they return (results for a CPU caller): after that synchronisation the flags of
        every pass of the call are on the host.  True = a network overflowed and has been switched to 3 x bf16 -- the caller repeats the
        work instead of handing out non-finite frames (and instead of reporting it one call late)."""
        if self._nf_event is None:
            return False
        self._nf_event.synchronize()
        bad = self._take_guard()
        if bad:
            import warnings
            warnings.warn(" and ".join("embedder" if n == "E" else "extractor" for n in bad) + ": data-dependent overflow of the 2 x f16 operand split; "
                          "switched to the exact 3 x bf16 split and repeated the call")
        return bool(bad)

    def _gemm_planes_ok(self, rows: int, n: int) -> bool:
        """1x1 GEMM on operand planes (gemm_pl.hip): 2 x f16 arithmetic and enough 256-row x 192-column tiles for the 256 CUs"""
        return self.planes_gemm and self.use_split and self.arith == 2 and ((rows + 255) // 256) * ((n + 191) // 192) >= 200

    def _planes_split(self, x: Act) -> int:
        """K slices of the planes conv for few key frames (shape-only rule, never timing: a K split changes the summation order): the smallest
        divisor-like count of the C/16 chunks that gives the 256 CUs >= 192 workgroups, every slice at least two chunks; 0 = not applicable"""
        tiles = x.B * (x.H // 16) * (x.W // 16) * (x.C // 192)
        if tiles >= 200:
            return 1
        if not self.planes_splitk:
            return 0
        spt = x.C // 16
        force = int(os.environ.get("VIDEOSEAL_PLANES_SK", "0"))        # experiments only (tools/r06_sk.sh): another slice count for launches below 200 tiles
        if force > 1 and spt % force == 0:
            return force
        for sk in (2, 3, 4, 6, 8, 12):
            cps = (spt + sk - 1) // sk
            if cps >= 2 and (sk - 1) * cps < spt and tiles * sk >= 192:
                return sk
        return 0

    def _planes_ok(self, x: Act, blocks) -> bool:
        """the bottleneck chain on pre-split operand planes (conv3x3_pl.hip): 2 x f16 arithmetic, eval BatchNorm folded, whole
        16 x 16-pixel tiles, 192-channel column tiles, and enough 256-pixel tiles (x K slices) to give (nearly) every CU a workgroup"""
        return (self.planes_chain and self.use_split and self.arith == 2 and len(blocks) > 0 and x.ld == x.C and x.C % 192 == 0 and
                x.H % 16 == 0 and x.W % 16 == 0 and self._planes_split(x) >= 1 and
                all("bn" not in p and "rms" not in p and p["cout"] == x.C and p["c0"].CinP == x.C and p["res"].CinP == x.C for p in blocks))

    def bottleneck_planes(self, x: Act, blocks, last_out: Optional[Act], xpl: Optional[torch.Tensor] = None) -> Act:
        """ResnetBlocks `blocks` (unet.py:24-39) from the fp32 activation x, every intermediate tensor as f16 operand planes: the
        activations are split once by the epilogue that produces them instead of by every consumer, and every conv runs on the all-DMA
        kernel (tile code 22).  Only the last block writes fp32 (into last_out's columns [0, C) if given)."""
        B, H, W, C = x.B, x.H, x.W, x.C
        self.planes_chain_ran = True                      # (bench.py names the dominant kernel after it)
        tile = N.CONV_TILE_HI | 6
        sk = self._planes_split(x)
        kw = dict(split_k=sk) if sk > 1 else {}
        ghost = Act(None, B, H, W, C, C)                  # geometry only: the tensor exists as planes
        if xpl is None:                                   # (the first block may hand its output over as planes already, resblock_msg0)
            xpl = self.to_planes(x, "bott.pl0")
        tpl = self.buf("bott.plt", x.rows * C).view(torch.int16)
        ypl = self.buf("bott.pl1", x.rows * C).view(torch.int16)
        out = None
        for j, p in enumerate(blocks):
            last = j == len(blocks) - 1
            self.conv(ghost, p["c0"], ghost, pad=1, act=N.ACT_RELU, tile_hint=tile, in_pl=xpl, out_pl=tpl, prof="bott.conv3x3", **kw)
            if last:
                out = last_out if last_out is not None else self.new_act("bott.o", B, H, W, C)
                self.conv(ghost, p["c1"], out, pad=1, act=N.ACT_RELU, in2=ghost, w2=p["res"], tile_hint=tile, in_pl=tpl, in2_pl=xpl,
                          n_store=(C if out.ld != rup(C, 4) else None), **kw)
            else:
                self.conv(ghost, p["c1"], ghost, pad=1, act=N.ACT_RELU, in2=ghost, w2=p["res"], tile_hint=tile, in_pl=tpl, in2_pl=xpl,
                          out_pl=ypl, **kw)
                xpl, ypl = ypl, xpl
        return out

    # ------------------------------------------------------------------ embedder
    def embedder_forward(self, x: Act, msgs_i32: torch.Tensor, bn_train: bool = False) -> torch.Tensor:
        """x: key frames, NHWC(ld 4), already mapped to [-1,1]. Returns delta [B][out_ch][S_h][S_w] (planar).
        bn_train: BatchNorm on batch statistics (module in .train() mode), running statistics updated in place."""
        for _ in range(2):
            self.arith = self.arith_net["E"]
            delta = self._embedder_forward(x, msgs_i32, bn_train)
            if self._guard("E", delta):
                return delta
            if bn_train and self.cfg.unet_norm != "rms":
                raise N.NativeError("embedder: non-finite output of a train-mode forward (BatchNorm statistics already updated) -- " + self._WHY)
        return delta

    def _embedder_forward(self, x: Act, msgs_i32: torch.Tensor, bn_train: bool = False) -> torch.Tensor:
        if self.cfg.unet_norm == "rms":
            bn_train = False              # no BatchNorm in the net: train and eval forwards coincide
        if bn_train:
            if self.Et is None:
                self._pack_embedder(self._g, train=True)
        elif self.E is None:
            self._pack_embedder(self._g)
        c, E, L = self.cfg, (self.Et if bn_train else self.E), self.lib
        B = x.B
        st = N.stream()
        rms = c.unet_norm == "rms"
        up_act = N.ACT_SILU if c.unet_act == "silu" else N.ACT_RELU        # unet.py:61-62: Upsample gets the U-Net's act_layer
        hid: List[Act] = [self.resblock(x, E["inc"], "inc")]
        nlev = len(c.zc) - 1
        for i in range(nlev):
            src = hid[-1]
            Ho, Wo = (src.H - 1) // 2 + 1, (src.W - 1) // 2 + 1
            dwn = self.new_act(f"down{i}.d", B, Ho, Wo, c.zc[i + 1])
            self.conv(src, E["downs"][i]["down"], dwn, stride=2, pad=1)
            if i == nlev - 1:     # last level lands in the message-augmented latent [lat | msg]
                h3 = self.new_act("h3", B, Ho, Wo, c.bott)
                if rms:           # the RMSNorm epilogue writes whole rows: land in a dense map, then copy into columns [0, zc[-1])
                    lat_ = self.resblock(dwn, E["downs"][i]["rb"], f"down{i}")
                    N.check(L.vs_cat2_scale(None, 0, 0, N.ptr(lat_.t), lat_.C, lat_.ld, 1.0, lat_.rows, N.ptr(h3.t), h3.ld, st), "vs_cat2_scale")
                else:
                    self.resblock(dwn, E["downs"][i]["rb"], f"down{i}", out=h3)
                hid.append(h3)
            else:
                hid.append(self.resblock(dwn, E["downs"][i]["rb"], f"down{i}"))
        h3 = hid[-1]
        Bm = msgs_i32.shape[0]
        lat = self.buf("msg.lat", Bm * c.hidden)
        N.check(L.vs_msg_latent(N.ptr(E["table"]), N.ptr(msgs_i32), Bm, c.nbits, c.hidden, N.ptr(lat), st), "vs_msg_latent")
        N.check(L.vs_broadcast_channels(N.ptr(lat), Bm, c.hidden, N.ptr(h3.t), B, h3.H * h3.W, h3.ld, c.zc[-1], st),
                "vs_broadcast_channels")
        xcur = h3
        def fused_ok(k: int, c1: int, c2: int) -> bool:
            up_ = E["ups"][k]
            return bool(self.upconv_fused and self.use_split and "gemm" in up_ and L.vs_upconv_fused_preferred(c1, c2, up_["gemm"].N // 9)
                        and up_["gemm"].CinP == c1 + c2)

        def lowres_cat(k: int, like: Act) -> Optional[Act]:
            """[x | skip] of Upsample group k at the LOW resolution; the producer of x writes columns [0, C) itself (eval mode)"""
            if k >= nlev or "gemm" not in E["ups"][k] or bn_train or rms or fused_ok(k, like.C, hid[nlev - k].C):
                return None
            return self.new_act(f"up{k}.lcat", B, like.H, like.W, like.C + hid[nlev - k].C)

        xcur_pl = None
        for j in range(c.num_blocks):
            lc = lowres_cat(0, xcur) if j == c.num_blocks - 1 else None
            if j == 0 and lc is None and self._msg0_ok(h3, E["bott"][0], c.zc[-1]):
                nxt = Act(None, B, h3.H, h3.W, E["bott"][0]["cout"], E["bott"][0]["cout"])
                chain = c.num_blocks > 1 and not bn_train and self._planes_ok(nxt, E["bott"][1:])
                r0 = self.resblock_msg0(h3, E["bott"][0], "bott0", lat, Bm, c.zc[-1], planes_out=chain)
                xcur, xcur_pl = r0 if chain else (r0, None)
                continue
            if j >= 1 and not bn_train and self._planes_ok(xcur, E["bott"][j:]):      # the rest of the chain on operand planes
                lc = lowres_cat(0, xcur)
                xcur = self.bottleneck_planes(xcur, E["bott"][j:], Act(lc.t, B, lc.H, lc.W, xcur.C, lc.ld) if lc else None, xpl=xcur_pl)
                break
            xcur = self.resblock(xcur, E["bott"][j], f"bott{j & 1}", out=(Act(lc.t, B, lc.H, lc.W, xcur.C, lc.ld) if lc else None))
        for k in range(nlev):
            skip = hid.pop()
            up = E["ups"][k]
            if "gemm" in up and xcur.ld == xcur.C and fused_ok(k, xcur.C, skip.C):     # thin levels: one kernel, z stays in LDS
                co = up["gemm"].N // 9
                ln = self.new_act(f"up{k}.ln", B, 2 * xcur.H, 2 * xcur.W, co)
                gw = up["gemm"].with_split(self.arith)
                N.check(L.vs_upconv_fused(N.ptr(xcur.t), xcur.C, xcur.ld, N.ptr(skip.t), skip.C, skip.ld, 2 ** -0.5,
                                          N.ptr(gw.split), B, xcur.H, xcur.W, co, N.ptr(up["lnw"]), N.ptr(up["lnb"]),
                                          1e-6, up_act, N.ptr(ln.t), ln.ld, self.arith, A_MUL, 1.0 / (A_MUL * gw.w_mul), st), "vs_upconv_fused")
            elif "gemm" in up:     # low-resolution 9-tap GEMM + gather / LayerNorm / ReLU (see vs_upconv_gather_ln)
                co = up["gemm"].N // 9
                direct = xcur.ld == xcur.C + skip.C      # x already sits in columns [0, C) of the concat buffer
                lc = Act(xcur.t, B, xcur.H, xcur.W, xcur.C + skip.C, xcur.ld) if direct else \
                    self.new_act(f"up{k}.lcat", B, xcur.H, xcur.W, xcur.C + skip.C)
                N.check(L.vs_cat2_scale(None if direct else N.ptr(xcur.t), xcur.C, xcur.ld, N.ptr(skip.t), skip.C, skip.ld, 2 ** -0.5,
                                        lc.rows, N.ptr(lc.t), lc.ld, st), "vs_cat2_scale")
                z = self.new_act(f"up{k}.z", B, lc.H, lc.W, 9 * co)
                self.conv(lc, up["gemm"], z, prof=(f"up{k}.gemm9" if self.time_all_convs else None))
                ln = self.new_act(f"up{k}.ln", B, 2 * lc.H, 2 * lc.W, co)
                N.check(L.vs_upconv_gather_ln(N.ptr(z.t), z.ld, B, lc.H, lc.W, co, N.ptr(up["lnw"]), N.ptr(up["lnb"]), 1e-6, up_act,
                                              N.ptr(ln.t), ln.ld, st), "vs_upconv_gather_ln")
            else:
                cat = self.new_act(f"up{k}.cat", B, 2 * xcur.H, 2 * xcur.W, xcur.C + skip.C)
                N.check(L.vs_upcat2x(N.ptr(xcur.t), xcur.C, xcur.ld, N.ptr(skip.t), skip.C, skip.ld, 2 ** -0.5, B, xcur.H, xcur.W,
                                     N.ptr(cat.t), cat.ld, st), "vs_upcat2x")
                cv = self.new_act(f"up{k}.conv", B, cat.H, cat.W, up["conv"].N)
                self.conv(cat, up["conv"], cv, pad=1, pad_mode=N.PAD_REFLECT)
                ln = self.new_act(f"up{k}.ln", B, cat.H, cat.W, up["conv"].N)
                self.layernorm(cv, up["lnw"], up["lnb"], ln, act=up_act)
            lc = lowres_cat(k + 1, ln)
            xcur = self.resblock(ln, up["rb"], f"up{k}", out=(Act(lc.t, B, lc.H, lc.W, up["rb"]["cout"], lc.ld) if lc else None))
        delta = self.buf("delta", B * c.out_ch * xcur.H * xcur.W)
        N.check(L.vs_outc_tanh(N.ptr(xcur.t), xcur.H * xcur.W, B, xcur.C, xcur.ld, N.ptr(E["outc_w"]), N.ptr(E["outc_b"]), c.out_ch,
                               1 if c.last_tanh else 0, N.ptr(delta), st), "vs_outc_tanh")
        return delta.view(B, c.out_ch, xcur.H, xcur.W)

    # ------------------------------------------------------------------ extractor
    def extractor_forward(self, x: Act) -> torch.Tensor:
        """x: NHWC(ld 4) RGB already mapped to [-1,1]. Returns logits [B][1+nbits]."""
        for attempt in range(3):
            self.arith = self.arith_net["X"]
            logits = self.vit_extractor_forward(x) if self.cfg.extractor == "sam" else self._extractor_forward(x)
            if self._guard("X", logits):
                break
            # the guard tripped on new weights and switched the network to 3 x bf16
            if self.layer_arith:                 # ... in the mixed configuration: it was calibrated on other data -- whole network exact
                self.layer_arith.clear()
            elif attempt == 0 and self.per_layer_arith and self.cfg.extractor != "sam":
                self._calibrate_extractor(x)     # may put the network back on the fast split with the overflowing layers pinned to 3 x bf16
        return logits

    def _note_absmax(self, key: tuple, t: torch.Tensor, numel: int) -> None:
        if self._calib is None:
            return
        slot = self._calib.setdefault(key, len(self._calib))
        N.check(self.lib.vs_absmax(N.ptr(t), numel, self._calib_bits.data_ptr() + 4 * slot, N.stream()), "vs_absmax")

    def _calibrate_extractor(self, x: Act) -> None:
        """one pass of the ConvNeXt extractor on the exact split that records max |operand| of every block GEMM, then the per-layer choice"""
        self._calib, self._calib_bits = {}, torch.zeros(4096, dtype=torch.int32, device=self.dev)
        try:
            self.arith = 3
            self._extractor_forward(x)
            bits = self._calib_bits.cpu()        # (one synchronisation per set of weights, like the guard's first read)
            keys = dict(self._calib)
        finally:
            self._calib = self._calib_bits = None
        amax = {k: float(bits[slot:slot + 1].view(torch.float32)) for k, slot in keys.items()}
        limit = 65504.0 / 4.0                    # 4 x head-room: the calibration saw one batch only
        over = {k for k, v in amax.items() if not (v * (A_MUL_GRN if k[-1] == "pw2" else A_MUL) < limit)}
        self.calib_absmax = amax
        if over and len(over) < len(amax):
            self.layer_arith = {k: 3 for k in over}
            self.arith_net["X"], self.verified["X"] = 2, False       # verified by the next pass of extractor_forward's loop
            import warnings
            warnings.warn(f"extractor: {len(over)} of {len(amax)} block GEMMs keep the exact 3 x bf16 split (operands beyond the f16 range of the "
                          f"2 x f16 split: {sorted(over)[:6]}{' ...' if len(over) > 6 else ''}); the other layers stay on the fast split")

    def _extractor_forward(self, x: Act) -> torch.Tensor:
        if self.X is None:
            self._pack_extractor(self._g)
        c, X, L = self.cfg, self.X, self.lib
        st = N.stream()
        B = x.B
        s = c.stem_stride
        Ho, Wo = (x.H - 4) // s + 1, (x.W - 4) // s + 1
        d = c.dims
        cur = self.new_act("st0.x", B, Ho, Wo, d[0], self._xld(d[0]))
        rc = N.ERR_UNSUPPORTED
        if self.stem_fused and x.ld == 4 and cur.ld == d[0] and X["stem"].bias is not None:
            # round 6: patchify conv + LayerNorm in one kernel on the vector ALUs (exact fp32 FMAs; 65 -> ~25 us for 32 frames)
            rc = L.vs_stem_conv_ln(N.ptr(x.t), B, x.H, x.W, s, N.ptr(X["stem"].wt), N.ptr(X["stem"].bias), N.ptr(X["stem_ln"][0]), N.ptr(X["stem_ln"][1]),
                                   1e-6, d[0], N.ptr(cur.t), cur.ld, st)
            if rc != N.ERR_UNSUPPORTED:
                N.check(rc, "vs_stem_conv_ln")
        if rc == N.ERR_UNSUPPORTED:
            t = self.new_act("stem.c", B, Ho, Wo, d[0])
            self.conv(x, X["stem"], t, geom=(Wo, s * 4, 16, s, 1, 0, 0))
            self.layernorm(t, X["stem_ln"][0], X["stem_ln"][1], cur)
        for sti in range(4):
            if sti > 0:
                dn = X["down"][sti - 1]
                Ho, Wo = cur.H // 2, cur.W // 2
                nxt = self.new_act(f"st{sti}.x", B, Ho, Wo, d[sti], self._xld(d[sti]))
                done = False
                if self.down_patch and cur.ld == cur.C and cur.C % 8 == 0 and dn["gemm"].CinP == 4 * cur.C and Ho > 0 and Wo > 0:
                    # round 6: LayerNorm straight into the patch matrix [B][H/2][W/2][4C] of the 2 x 2 / stride-2 conv, which then is a dense 1x1
                    # GEMM (the wave-specialised kernel by the library's rule) instead of the generic strided conv: 72 -> 3x us at 96 -> 192 @32^2
                    lnp = Act(self.buf(f"st{sti}.dlnp", B * Ho * Wo * 4 * cur.C), B, Ho, Wo, 4 * cur.C, 4 * cur.C)
                    rc = L.vs_layernorm_patch2x2(N.ptr(cur.t), B, cur.H, cur.W, cur.C, cur.ld, N.ptr(dn["lnw"]), N.ptr(dn["lnb"]), 1e-6, N.ptr(lnp.t), st)
                    if rc != N.ERR_UNSUPPORTED:
                        N.check(rc, "vs_layernorm_patch2x2")
                        self.conv(lnp, dn["gemm"], nxt)
                        done = True
                if not done:
                    ln = self.new_act(f"st{sti}.dln", B, cur.H, cur.W, cur.C, cur.ld)
                    self.layernorm(cur, dn["lnw"], dn["lnb"], ln)
                    self.conv(ln, dn["conv"], nxt, geom=(Wo, 2 * ln.ld, 2 * ln.ld, 2, 1, 0, 0))
                cur = nxt
            Cc = d[sti]
            HW = cur.H * cur.W
            tn = self.new_act(f"st{sti}.n", B, cur.H, cur.W, Cc, self._xld(Cc))
            hh = self.new_act(f"st{sti}.h", B, cur.H, cur.W, 4 * Cc, self._xld(4 * Cc))
            nchunk = (HW + 63) // 64
            part = self.buf(f"st{sti}.gp", nchunk * B * 4 * Cc)
            scale = self.buf(f"st{sti}.gs", B * hh.ld + 16)   # +16: the conv A-transform reads whole 16-float chunks
            # all-DMA GEMM on operand planes (gemm_pl.hip, tile code 24) where its 256-row tiles give every CU work: pwconv1 reads the
            # LayerNorm output as f16 planes written by the dwconv kernel itself; pwconv2 only for long K (K >= 2048: the GRN apply then
            # runs in a separate conversion pass, which costs more than it saves on the shorter layers -- tools/bench_gemm.py planes)
            pw1w = X["stages"][sti][0]["pw1"]
            pl1 = self._gemm_planes_ok(cur.rows, 4 * Cc) and pw1w.CinP % 16 == 0
            pl2, sk2 = False, 1
            if self.planes_gemm and self.use_split and self.arith == 2 and hh.ld >= 2048 and hh.ld % 16 == 0:
                tiles2 = ((cur.rows + 255) // 256) * ((Cc + 191) // 192)
                steps2 = hh.ld // 16
                while tiles2 * sk2 < 200 and steps2 // (sk2 * 2) >= 24 and steps2 % (sk2 * 2) == 0:
                    sk2 *= 2
                pl2 = tiles2 * sk2 >= 128
                # round 5: small-M layers (VideoSeal stage 3: 2048 rows x K = 3072) go back to the wave-specialised GEMM with the GRN transform on
                # its A path -- with the producers' lanes all busy it is 50.9 us (4 K slices) against 15.0 (conversion pass over h) + 51.5 us on
                # planes (tools/bench_gemm.py planes); ChunkySeal's wide layers (thousands of rows, K slices beyond the GRN rows' LDS budget) stay
                if pl2 and cur.rows <= 2048 and HW % 64 == 0 and hh.ld % 32 == 0 and self.pw2_small_pc:
                    skp = 1
                    blocks = ((cur.rows + 127) // 128) * ((Cc + 127) // 128)
                    pairs = hh.ld // 32
                    while blocks * skp < 256 and pairs % (skp * 2) == 0 and pairs // (skp * 2) >= 4:
                        skp *= 2
                    if hh.ld // skp <= 3072:
                        pl2 = False
            elif (self.planes_gemm and self.use_split and self.arith == 2 and hh.ld % 16 == 0 and HW % 64 != 0
                  and ((cur.rows + 255) // 256) * ((Cc + 191) // 192) >= 200):
                # round 6 (ChunkySeal stages 0 / 1: 127 x 127 and 63 x 63 frames, K = 1448 / 2896 < 2048 at stage 0): frames that do not align with the
                # wave-specialised GEMM's 64-row halves pay a full pass over h anyway (vs_grn_apply in place) and then run the slower kernel -- the
                # planes conversion IS that pass, and the all-DMA GEMM behind it is twice as fast (1.6 -> 0.85 ms per launch at 258 064 x 1448 x 362)
                pl2, sk2 = True, 1
            tnpl = self.buf(f"st{sti}.npl", cur.rows * pw1w.CinP).view(torch.int16) if pl1 else None
            hpl = self.buf(f"st{sti}.hpl", cur.rows * hh.ld).view(torch.int16) if pl2 else None
            ptile = N.CONV_TILE_HI | 8
            # round 6: 256 x 256 tiles with one wave per SIMD (tile 27) for the GEMMs with several rounds of 256 x 192 tiles per launch -- those run at the
            # chip's LDS-DMA stream rate, and the larger tile moves 15 % fewer operand bytes per FLOP (ChunkySeal; VIDEOSEAL_GEMM_BIG=0 keeps tile 24)
            big1 = self.gemm_big and pl1 and ((cur.rows + 255) // 256) * ((4 * Cc + 255) // 256) >= 3 * 256
            big2 = self.gemm_big and pl2 and sk2 == 1 and ((cur.rows + 255) // 256) * ((Cc + 255) // 256) >= 3 * 256
            ptile1 = (N.CONV_TILE_HI | 11) if big1 else ptile
            ptile2 = (N.CONV_TILE_HI | 11) if big2 else ptile
            fused = (self.fused_blocks and self.arith == 2 and X["stages"][sti] and X["stages"][sti][0].get("fuse") is not None
                     and pw1w.CinP == Cc and bool(L.vs_cnx_block_supported(Cc, cur.rows, HW)))
            if fused and tnpl is None:
                tnpl = self.buf(f"st{sti}.npl", cur.rows * pw1w.CinP).view(torch.int16)
            calib = self._calib is not None
            for bj, blk in enumerate(X["stages"][sti]):
                # arithmetic of this block's two GEMMs: the network's, unless the calibration pinned the layer to the exact split
                a1 = 3 if (self.arith == 3 or self.layer_arith.get((sti, bj, "pw1")) == 3) else 2
                a2 = 3 if (self.arith == 3 or self.layer_arith.get((sti, bj, "pw2")) == 3) else 2
                mixed = self.arith == 2 and (a1 == 3 or a2 == 3)
                kwa1, kwa2 = (dict(arith=a1) if mixed else {}), (dict(arith=a2) if mixed else {})
                if fused and a1 == 2 and a2 == 2:    # pwconv1 -> GELU -> GRN -> pwconv2 with h on chip: statistics pass, scale, apply pass (in place on cur)
                    img, m1, m2 = blk["fuse"]
                    N.check(L.vs_dwconv7_ln_planes(N.ptr(cur.t), B, cur.H, cur.W, Cc, cur.ld, N.ptr(blk["wdw"]), N.ptr(blk["bdw"]), N.ptr(blk["lnw"]),
                                                   N.ptr(blk["lnb"]), 1e-6, A_MUL, pw1w.CinP, N.ptr(tnpl), st), "vs_dwconv7_ln_planes")
                    part32 = self.buf(f"st{sti}.gp32", B * (HW // 32) * 4 * Cc)
                    am1, am2 = 1.0 / (A_MUL * m1), 1.0 / (A_MUL_GRN * m2)
                    N.check(L.vs_cnx_block(N.ptr(tnpl), N.ptr(img), Cc, cur.rows, HW, 1, am1, am2, None, 0, None, None, 0, None, 0, N.ptr(part32), st),
                            "vs_cnx_block(stats)")
                    N.check(L.vs_grn_scale_from_partials(N.ptr(part32), B, HW, 4 * Cc, N.ptr(blk["gamma"]), N.ptr(scale), hh.ld, st),
                            "vs_grn_scale_from_partials")
                    N.check(L.vs_cnx_block(N.ptr(tnpl), N.ptr(img), Cc, cur.rows, HW, 0, am1, am2, N.ptr(scale), hh.ld, N.ptr(blk["pw2"].bias),
                                           N.ptr(cur.t), cur.ld, N.ptr(cur.t), cur.ld, None, st), "vs_cnx_block(apply)")
                    continue
                bpl1, bpl2 = pl1 and a1 == 2, pl2 and a2 == 2
                if bpl1:
                    N.check(L.vs_dwconv7_ln_planes(N.ptr(cur.t), B, cur.H, cur.W, Cc, cur.ld, N.ptr(blk["wdw"]), N.ptr(blk["bdw"]), N.ptr(blk["lnw"]),
                                                   N.ptr(blk["lnb"]), 1e-6, A_MUL, pw1w.CinP, N.ptr(tnpl), st), "vs_dwconv7_ln_planes")
                else:
                    N.check(L.vs_dwconv7_ln(N.ptr(cur.t), B, cur.H, cur.W, Cc, cur.ld, N.ptr(blk["wdw"]), N.ptr(blk["bdw"]), N.ptr(blk["lnw"]),
                                            N.ptr(blk["lnb"]), 1e-6, N.ptr(tn.t), tn.ld, st), "vs_dwconv7_ln")
                    self._note_absmax((sti, bj, "pw1"), tn.t, tn.rows * tn.ld)
                kw1 = dict(in_pl=tnpl, tile_hint=ptile1) if bpl1 else dict(kwa1)
                if self.prof_extractor and sti == 2:      # bench.py --detect-only: the dominant GEMM of an extractor-only workload
                    kw1["prof"] = (f"{'gemm_pl_kernel' if bpl1 else 'gemm1x1_pc_kernel / conv_gemm_kernel'}: ConvNeXt stage-2 pwconv1 "
                                   f"{Cc}->{4 * Cc} @{cur.H}x{cur.W}")
                    kw1["flops"] = 2.0 * cur.rows * Cc * 4 * Cc      # algorithmic (unpadded) FLOPs
                fold = None
                if HW % 32 == 0:      # ||h||^2 partials come out of pwconv1's epilogue: no second pass over h
                    part32 = self.buf(f"st{sti}.gp32", B * (HW // 32) * 4 * Cc)
                    self.conv(tn, blk["pw1"], hh, act=N.ACT_GELU, sumsq=part32, **kw1)
                    # the finish (partials -> scale) is deferred to pwconv2's launch: conv(grn_fold=...) folds it into the wave-specialised GEMM's
                    # prologue where that kernel runs (round 6: 12 launches fewer per extractor pass), and issues it as its own launch otherwise
                    fold = (part32, blk["gamma"], B, HW, 4 * Cc)
                    if not ((HW % 64 == 0 or not self.use_split) and not calib and not (pl2 and a2 == 2)):
                        N.check(L.vs_grn_scale_from_partials(N.ptr(part32), B, HW, 4 * Cc, N.ptr(blk["gamma"]), N.ptr(scale), hh.ld, st),
                                "vs_grn_scale_from_partials")
                        fold = None
                elif bpl1 and HW >= 32 and self.grn_straddle:
                    # round 6 (ChunkySeal: 31 x 31 frames): the planes GEMM's epilogue writes the partial sums of its 32-row groups split at the frame
                    # boundary a group may straddle -- the separate pass over h (356 MB per block at ChunkySeal's size) is gone
                    parts = self.buf(f"st{sti}.gps", ((cur.rows + 31) // 32) * 2 * 4 * Cc)
                    self.conv(tn, blk["pw1"], hh, act=N.ACT_GELU, sumsq=parts, sumsq_hw=HW, **kw1)
                    N.check(L.vs_grn_scale_from_straddle_partials(N.ptr(parts), B, HW, 4 * Cc, N.ptr(blk["gamma"]), N.ptr(scale), hh.ld, st),
                            "vs_grn_scale_from_straddle_partials")
                else:
                    self.conv(tn, blk["pw1"], hh, act=N.ACT_GELU, **kw1)
                    N.check(L.vs_grn_scale(N.ptr(hh.t), B, HW, 4 * Cc, hh.ld, N.ptr(blk["gamma"]), N.ptr(part), N.ptr(scale), st),
                            "vs_grn_scale")
                if bpl2:              # GRN apply + operand split in one pass over h, then the planes GEMM (K split by the shape rule)
                    N.check(L.vs_to_planes_affine(N.ptr(hh.t), hh.rows, hh.ld, hh.ld, A_MUL_GRN, N.ptr(scale), hh.ld, N.ptr(blk["beta"]), HW,
                                                  N.ptr(hpl), st), "vs_to_planes_affine")
                    self.conv(hh, blk["pw2"], cur, res=cur, in_pl=hpl, tile_hint=ptile2, split_k=sk2, a_mul=A_MUL_GRN)
                elif (HW % 64 == 0 or not self.use_split) and not calib:
                    kn = dict(kwa2)
                    # round 5: where 128-row x 128-column tiles would need K slices to occupy the CUs (stage 2: 64 x 3 tiles -> 2 slices + an
                    # epilogue launch that adds 25 MB of partial sums), 128 x 96 tiles give 64 x 4 workgroups with the whole K each: one launch
                    # (tile 26 takes the whole K in one workgroup: its GRN rows must fit the kernel's LDS area, K <= 3072, whole K pairs)
                    if (self.pw2_narrow and self.use_split and a2 == 2 and ((cur.rows + 127) // 128) * ((Cc + 127) // 128) < 256
                            and ((cur.rows + 127) // 128) * ((Cc + 95) // 96) >= 200 and Cc % 96 == 0 and HW >= 128
                            and hh.ld <= 3072 and hh.ld % 32 == 0 and hh.ld == 4 * Cc):
                        kn.update(tile_hint=N.CONV_TILE_HI | 10, split_k=1)
                    self.conv(hh, blk["pw2"], cur, res=cur, a_scale=scale, a_scale_ld=hh.ld, a_shift=blk["beta"], grn_fold=fold, **kn)
                else:     # odd feature maps (ChunkySeal: 31 x 31): GRN applied in place + plain GEMM measured faster (109 vs 105 frames/s)
                          # than the GEMM with the fused transform, whose frame-boundary select costs registers; the calibration pass takes
                          # this form everywhere because it measures the operand h * scale + beta itself
                    N.check(L.vs_grn_apply(N.ptr(hh.t), B, HW, 4 * Cc, hh.ld, N.ptr(scale), hh.ld, N.ptr(blk["beta"]), st), "vs_grn_apply")
                    self._note_absmax((sti, bj, "pw2"), hh.t, hh.rows * hh.ld)
                    self.conv(hh, blk["pw2"], cur, res=cur, a_mul=A_MUL_GRN, **kwa2)
        return self._pixel_decoder(cur, X)

    def _pixel_decoder(self, cur: Act, X) -> torch.Tensor:
        """pixel_decoder.py:61-83 with upscale_stages [1]: reflect-pad conv3x3 -> LayerNorm(cf) -> GELU -> mean(H, W) -> Linear."""
        c, L, st, B = self.cfg, self.lib, N.stream(), cur.B
        d = [cur.C]
        hc = self.new_act("head.c", B, cur.H, cur.W, d[-1])
        hw = X["head_conv"]
        if cur.rows <= 4096 and cur.H > 1 and cur.W > 1 and hw.CinP == cur.ld and self.use_split:
            # few rows, long K (VideoSeal: 2048 x 6912): patch matrix + dense split-K GEMM instead of 96 implicit-GEMM workgroups
            if "head_gemm" not in X:
                X["head_gemm"] = ConvW(hw.wt, None, hw.N, 1, 1, 9 * hw.CinP)
            cols = Act(self.buf("head.cols", cur.rows * 9 * cur.ld), B, cur.H, cur.W, 9 * cur.ld, 9 * cur.ld)
            N.check(L.vs_im2col3x3(N.ptr(cur.t), B, cur.H, cur.W, cur.ld, N.PAD_REFLECT, N.ptr(cols.t), st), "vs_im2col3x3")
            self.conv(cols, X["head_gemm"], hc)
        else:
            self.conv(cur, hw, hc, pad=1, pad_mode=N.PAD_REFLECT)
        hl = self.new_act("head.l", B, cur.H, cur.W, d[-1])
        self.layernorm(hc, X["head_ln"][0], X["head_ln"][1], hl, act=N.ACT_GELU)
        out = self.buf("logits", B * (c.nbits + 1)).view(B, c.nbits + 1)
        N.check(L.vs_pool_linear(N.ptr(hl.t), B, hl.H * hl.W, hl.C, hl.ld, N.ptr(X["lin_w"]), N.ptr(X["lin_b"]), c.nbits + 1, N.ptr(out), st),
                "vs_pool_linear")
        return out

    # ------------------------------------------------------------------ ViT extractor (legacy videoseal_0.0 card)
    def _pack_vit(self, g):
        c = self.cfg
        ie = "detector.image_encoder"
        D_ = c.vit_dim
        V = {}

        def lin(key, in_ld):
            w, cp = pack_conv(g(key + ".weight").float()[:, :, None, None], in_ld)
            return ConvW(w, g(key + ".bias").float().contiguous(), w.shape[0], 1, 1, cp)

        wp, cpp = pack_patch_conv(g(ie + ".patch_embed.proj.weight"), 4)
        V["patch"] = ConvW(wp, g(ie + ".patch_embed.proj.bias").float().contiguous(), D_, c.vit_patch, 1, cpp)
        V["pos"] = g(ie + ".pos_embed").float().reshape(-1, D_).contiguous()          # [g*g][D]
        V["blocks"] = []
        for i in range(c.vit_depth):
            p = f"{ie}.blocks.{i}"
            V["blocks"].append(dict(
                n1=(g(p + ".norm1.weight").float().contiguous(), g(p + ".norm1.bias").float().contiguous()),
                n2=(g(p + ".norm2.weight").float().contiguous(), g(p + ".norm2.bias").float().contiguous()),
                qkv=lin(p + ".attn.qkv", D_), proj=lin(p + ".attn.proj", D_),
                lin1=lin(p + ".mlp.lin1", D_), lin2=lin(p + ".mlp.lin2", int(D_ * c.vit_mlp_ratio)),
                rel_h=g(p + ".attn.rel_pos_h").float().contiguous() if c.vit_rel_pos else None,
                rel_w=g(p + ".attn.rel_pos_w").float().contiguous() if c.vit_rel_pos else None,
                window=0 if i in c.vit_global else c.vit_window))
        wn0, cp0 = pack_conv(g(ie + ".neck.0.weight"), rup(D_, 4))
        V["neck0"] = ConvW(wn0, None, c.vit_out, 1, 1, cp0)
        V["neck1"] = (g(ie + ".neck.1.weight").float().contiguous(), g(ie + ".neck.1.bias").float().contiguous())
        wn2, cp2 = pack_conv(g(ie + ".neck.2.weight"), self._xld(c.vit_out))
        V["neck2"] = ConvW(wn2, None, c.vit_out, 3, 3, cp2)
        V["neck3"] = (g(ie + ".neck.3.weight").float().contiguous(), g(ie + ".neck.3.bias").float().contiguous())
        pd = "detector.pixel_decoder"
        wh, cph = pack_conv(g(pd + ".output_upscaling.0.upsample_block.2.weight"), self._xld(c.vit_out))
        V["head_conv"] = ConvW(wh, None, c.vit_out, 3, 3, cph)
        V["head_ln"] = (g(pd + ".output_upscaling.0.upsample_block.3.weight").float().contiguous(), g(pd + ".output_upscaling.0.upsample_block.3.bias").float().contiguous())
        V["lin_w"] = g(pd + ".linear.weight").float().contiguous()
        V["lin_b"] = g(pd + ".linear.bias").float().contiguous()
        self.X = V
        self._sd_now = None

    def vit_extractor_forward(self, x: Act) -> torch.Tensor:
        """extractor.py:45-75 + vit.py:129-144: x (NHWC ld 4, already x*2-1) -> patch embedding + absolute positions -> blocks
        (LayerNorm -> windowed / global attention with decomposed relative positions -> +x -> LayerNorm -> MLP -> +x) -> neck ->
        pixel decoder -> logits [B][1+nbits].  Linears and convs: vs_conv_gemm; attention: vs_vit_attention."""
        if self.X is None:
            self._pack_vit(self._g)
        c, V, L, st, B = self.cfg, self.X, self.lib, N.stream(), x.B
        P, D_ = c.vit_patch, c.vit_dim
        gh, gw = x.H // P, x.W // P
        if gh * gw * D_ != V["pos"].numel():
            raise N.NativeError(f"ViT extractor: {x.H}x{x.W} input does not match the {int(math.isqrt(V['pos'].shape[0])) * P}^2 position table")
        pos = self._ws.get(("vit.pos", B))
        if pos is None:                      # absolute positions as the patch conv's residual operand: one row per token of every frame
            pos = V["pos"].repeat(B, 1).contiguous()
            self._ws[("vit.pos", B)] = pos
        tok = self.new_act("vit.x", B, gh, gw, D_)
        self.conv(x, V["patch"], tok, geom=(gw, P * 4, P * 4, P, 1, 0, 0), res=Act(pos, B, gh, gw, D_, D_))
        hd = D_ // c.vit_heads
        hid = int(D_ * c.vit_mlp_ratio)
        nrm = self.new_act("vit.n", B, gh, gw, D_)
        qkv = self.new_act("vit.qkv", B, gh, gw, 3 * D_)
        att = self.new_act("vit.att", B, gh, gw, D_)
        mid = self.new_act("vit.mid", B, gh, gw, hid)
        for blk in V["blocks"]:
            self.layernorm(tok, blk["n1"][0], blk["n1"][1], nrm, eps=1e-5)           # nn.LayerNorm default eps
            self.conv(nrm, blk["qkv"], qkv)
            N.check(L.vs_vit_attention(N.ptr(qkv.t), B, gh, gw, c.vit_heads, hd, blk["window"], N.ptr(blk["rel_h"]), N.ptr(blk["rel_w"]),
                                       N.ptr(att.t), st), "vs_vit_attention")
            self.conv(att, blk["proj"], tok, res=tok)                                # x = shortcut + proj(attn)
            self.layernorm(tok, blk["n2"][0], blk["n2"][1], nrm, eps=1e-5)
            self.conv(nrm, blk["lin1"], mid, act=N.ACT_GELU)
            self.conv(mid, blk["lin2"], tok, res=tok)                                # x = x + mlp(norm2(x))
        n0 = self.new_act("vit.neck0", B, gh, gw, c.vit_out)
        self.conv(tok, V["neck0"], n0)
        n1 = self.new_act("vit.neck1", B, gh, gw, c.vit_out, self._xld(c.vit_out))
        self.layernorm(n0, V["neck1"][0], V["neck1"][1], n1)
        n2 = self.new_act("vit.neck2", B, gh, gw, c.vit_out)
        self.conv(n1, V["neck2"], n2, pad=1)
        n3 = self.new_act("vit.neck3", B, gh, gw, c.vit_out, self._xld(c.vit_out))
        self.layernorm(n2, V["neck3"][0], V["neck3"][1], n3)
        return self._pixel_decoder(n3, V)

    # ------------------------------------------------------------------ shell
    def resize_pre(self, imgs: torch.Tensor, S: Tuple[int, int], antialias: bool, *, want_rgb: bool, mul=1.0, add=0.0,
                   want_key: bool = False, key_step: int = 1, tag="rs") -> Tuple[Optional[Act], Optional[Act]]:
        """imgs on device, fp32 NCHW [B,3,H,W] or uint8 RGB24 [B,H,W,3] -> (rgb Act [B,S,S,4] or None, key Act [ceil(B/step),S,S,4] or None)."""
        u8 = imgs.dtype == torch.uint8
        if u8:
            B, H, W, Cc = imgs.shape
            if Cc != 3:
                raise ValueError("uint8 frames must be RGB24 [F, H, W, 3]")
        else:
            B, Cc, H, W = imgs.shape
        rgb = self.new_act(tag + ".rgb", B, S[0], S[1], 3, 4) if want_rgb else None
        nk = (B + key_step - 1) // key_step
        key = self.new_act(tag + ".key", nk, S[0], S[1], self.cfg.in_ch, 4) if want_key else None
        ymat = self.ymat if self.cfg.yuv else None
        ev = self._shell_t0()
        if u8:
            N.check(self.lib.vs_resize_pre_u8(N.ptr(imgs), B, H, W, S[0], S[1], 1 if antialias else 0, N.ptr(rgb.t) if rgb else None,
                                              mul, add, N.ptr(key.t) if key else None, key_step, ymat, N.stream()), "vs_resize_pre_u8")
        else:
            N.check(self.lib.vs_resize_pre(N.ptr(imgs), B, Cc, H, W, S[0], S[1], 1 if antialias else 0, N.ptr(rgb.t) if rgb else None,
                                           mul, add, N.ptr(key.t) if key else None, key_step, ymat, N.stream()), "vs_resize_pre")
        # algorithmic HBM bytes: the frame once + the 256^2 outputs (4 floats per pixel)
        self._shell_t1(ev, "resize_pre_kernel" + ("<u8>" if u8 else ""),
                       imgs.numel() * imgs.element_size() + 16 * S[0] * S[1] * ((B if rgb else 0) + (nk if key else 0)))
        return rgb, key

    def _shell_t0(self):
        if self.shell_timers is None or torch.cuda.is_current_stream_capturing():
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _shell_t1(self, ev0, name: str, nbytes: int):
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.shell_timers.append((name, ev0, ev1, nbytes))

    def jnd_lowres(self, rgb: Act) -> torch.Tensor:
        h = self.buf("jnd.low", rgb.B * rgb.H * rgb.W)
        N.check(self.lib.vs_jnd_heatmap(N.ptr(rgb.t), rgb.B, rgb.H, rgb.W, rgb.H * rgb.W * rgb.ld, 1, rgb.W * rgb.ld, rgb.ld,
                                        self.taps43, N.ptr(h), N.stream()), "vs_jnd_heatmap")
        return h

    def jnd_full(self, imgs: torch.Tensor) -> torch.Tensor:
        B, _, H, W = imgs.shape
        h = torch.empty(B, 1, H, W, device=self.dev, dtype=torch.float32)
        N.check(self.lib.vs_jnd_heatmap(N.ptr(imgs), B, H, W, 3 * H * W, H * W, W, 1, self.taps43, N.ptr(h), N.stream()),
                "vs_jnd_heatmap")
        return h

    def embed_tail(self, imgs, out, delta, *, step, video_mode, hmap_low, attenuate, clamp, antialias, scaling_i, scaling_w,
                   preds_w=None):
        d = N.TailDesc()
        u8 = imgs.dtype == torch.uint8          # RGB24 [F,H,W,3] in and out
        if u8:
            F_, H, W, _ = imgs.shape
            if out.dtype != torch.uint8 or out.shape != imgs.shape:
                raise ValueError("uint8 frames need a uint8 output of the same shape")
            d.io_u8 = 1
        else:
            F_, _, H, W = imgs.shape
        d.imgs, d.out, d.preds_w = N.ptr(imgs), N.ptr(out), N.ptr(preds_w)
        d.delta, d.hmap_lowres = N.ptr(delta), N.ptr(hmap_low)
        d.taps43 = C.cast(self.taps43, C.c_void_p)
        d.F, d.H, d.W, d.S_h, d.S_w, d.Cd = F_, H, W, delta.shape[-2], delta.shape[-1], delta.shape[1]
        d.step, d.video_mode, d.total_key = step, video_mode, delta.shape[0]
        d.attenuate, d.clamp, d.antialias = int(attenuate), int(clamp), int(antialias)
        d.scaling_i, d.scaling_w = float(scaling_i), float(scaling_w)
        d.variant = int(self.tail_variant)
        ev = self._shell_t0()
        N.check(self.lib.vs_embed_tail(C.byref(d), N.stream()), "vs_embed_tail")
        # algorithmic HBM bytes: frame read + watermarked frame written (+ preds_w); the 256^2 delta / heat-map reads are cache hits
        self._shell_t1(ev, "embed_tail_kernel" + ("<u8>" if u8 else ""), imgs.numel() * imgs.element_size() + out.numel() * out.element_size()
                       + (preds_w.numel() * 4 if preds_w is not None else 0) + delta.numel() * 4)
