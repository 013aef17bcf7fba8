"""Per-kernel parity: every C-ABI entry point against the same op computed by torch fp32 on the CPU
(ATen is what the reference runs).  Tolerances are written next to each check.  Needs an MI355X."""
import os
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from videoseal_amd import native as N  # noqa: E402
from videoseal_amd.engine import Act, ConvW, HipEngine, pack_conv, pack_patch_conv, rup  # noqa: E402

DEV = "cuda"


def to_nhwc(x, ld=None):
    B, Cc, H, W = x.shape
    ld = ld or rup(Cc, 4)
    t = torch.zeros(B, H, W, ld)
    t[..., :Cc] = x.permute(0, 2, 3, 1)
    return Act(t.to(DEV).contiguous(), B, H, W, Cc, ld)


def from_nhwc(a: Act, Cc=None):
    Cc = Cc or a.C
    return a.t.view(a.B, a.H, a.W, a.ld)[..., :Cc].permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("cfg", [(15, 2, True), (0x40, 3, True), (15, 4, False), (0x40, 6, False), (15, 8, True)])
def test_patch_pc_split_k(eng, cfg):
    """wave-specialised 3x3 kernel with the K loop cut into chunk slices (+ one slice for the fused 1x1 res_conv) and the
    shared split-K epilogue: relu(conv3x3(t)+b) + conv1x1(x)+b2 + res, written at a channel offset."""
    if not eng.use_split:
        pytest.skip("split back-end only")
    tile, sk, two = cfg
    g = torch.Generator().manual_seed(17)
    B, Cm, Cx, H, W, Co = 2, 384, 48, 16, 32, 200
    t = torch.randn(B, Cm, H, W, generator=g)
    x = torch.randn(B, Cx, H, W, generator=g)
    w1 = torch.randn(Co, Cm, 3, 3, generator=g) / math.sqrt(Cm * 9)
    b1 = torch.randn(Co, generator=g)
    w2 = torch.randn(Co, Cx, 1, 1, generator=g) / math.sqrt(Cx)
    b2 = torch.randn(Co, generator=g)
    res = torch.randn(B, Co, H, W, generator=g)
    ref = F.relu(F.conv2d(t, w1, b1, padding=1)) + res
    if two:
        ref = ref + F.conv2d(x, w2, b2)
    ta, xa, ra = to_nhwc(t), to_nhwc(x), to_nhwc(res)
    wt1, cp1 = pack_conv(w1.to(DEV), ta.ld)
    wt2, cp2 = pack_conv(w2.to(DEV), xa.ld)
    out = eng.new_act("wide2", B, H, W, 208)
    out.t.fill_(7.0)
    kw = dict(in2=xa, w2=ConvW(wt2, b2.to(DEV), Co, 1, 1, cp2)) if two else {}
    eng.conv(ta, ConvW(wt1, b1.to(DEV), Co, 3, 3, cp1), out, pad=1, act=N.ACT_RELU, res=ra, out_coff=4, n_store=Co, tile_hint=tile,
             split_k=sk, **kw)
    torch.cuda.synchronize()
    full = out.t.view(B, H, W, 208).cpu()
    assert rel_err(full[..., 4:204].permute(0, 3, 1, 2), ref) < 2e-5
    assert (full[..., :4] == 7.0).all() and (full[..., 204:] == 7.0).all()


@pytest.mark.parametrize("cfg", [(2, 32, 16, 16, 192, False), (2, 48, 32, 16, 384, True), (3, 64, 16, 48, 200, True), (1, 16, 16, 16, 64, False)])
def test_conv3x3_planes_kernel(cfg):
    """all-DMA 3x3 kernel on pre-split operand planes (tile codes 22 / 23, conv3x3_pl.hip): fp32 output bit-identical to the
    wave-specialised kernel in the same 2 x f16 arithmetic (same products, same K order), planes output == vs_to_planes of it,
    and both against torch."""
    B, C, H, W, Co, two = cfg
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(29)
    x = torch.randn(B, C, H, W, generator=g)
    x2 = torch.randn(B, C, H, W, generator=g)
    w1 = torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b1 = torch.randn(Co, generator=g)
    w2 = torch.randn(Co, C, 1, 1, generator=g) / math.sqrt(C)
    b2 = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x, w1, b1, padding=1))
    if two:
        ref = ref + F.conv2d(x2, w2, b2)
    xa, xa2 = to_nhwc(x), to_nhwc(x2)
    wt1, cp1 = pack_conv(w1.to(DEV), xa.ld)
    wt2, cp2 = pack_conv(w2.to(DEV), xa2.ld)
    cw1, cw2 = ConvW(wt1, b1.to(DEV), Co, 3, 3, cp1), ConvW(wt2, b2.to(DEV), Co, 1, 1, cp2)
    xpl, x2pl = eng.to_planes(xa, "t.xpl"), eng.to_planes(xa2, "t.x2pl")
    # planes layout [2][C/16][rows][16]: hi + lo == x * 16 up to the f16 split's 2^-24 relative error
    pl = xpl.view(torch.float16).float().view(2, C // 16, B * H * W, 16).sum(0).permute(1, 0, 2).reshape(B, H, W, C).cpu() / 16.0
    assert (pl - x.permute(0, 2, 3, 1)).abs().max() < 1e-6
    outs = []
    opl = eng.buf("t.opl", B * H * W * rup(Co, 16)).view(torch.int16)
    for tile in (N.CONV_TILE_HI | 6, N.CONV_TILE_HI | 7, 15):
        out = eng.new_act(f"t.plo{tile}", B, H, W, Co)
        out.t.fill_(3.0)
        kw = dict(in2=xa2, w2=cw2) if two else {}
        if tile != 15:
            kw.update(in_pl=xpl, in2_pl=(x2pl if two else None), out_pl=(opl if Co % 16 == 0 else None))
        eng.conv(xa, cw1, out, pad=1, act=N.ACT_RELU, tile_hint=tile, arith=2, **kw)
        torch.cuda.synchronize()
        outs.append(out.t.clone())
        assert rel_err(from_nhwc(out), ref) < 2e-5
        if tile != 15 and Co % 16 == 0:
            want = eng.to_planes(Act(out.t, B, H, W, Co, out.ld), "t.wpl")
            torch.cuda.synchronize()
            assert torch.equal(opl[: want.numel()], want)
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("cfg", [(2, 96, 16, 16, 192, True, 2), (2, 96, 16, 16, 192, True, 3), (1, 384, 32, 32, 384, True, 6), (1, 384, 16, 32, 384, False, 8),
                                 (2, 64, 16, 16, 64, True, 2)])
def test_conv3x3_planes_kernel_with_k_slices(cfg):
    """the planes kernel with K slices (few output tiles): partial sums per slice + split-K epilogue (bias, ReLU, the 1x1 phase as its own
    slice, fp32 and / or planes output) against the unsplit launch (same values up to the summation order) and torch; deterministic"""
    B, C, H, W, Co, two, sk = cfg
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, H, W, generator=g)
    x2 = torch.randn(B, C, H, W, generator=g)
    w1 = torch.randn(Co, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b1 = torch.randn(Co, generator=g)
    w2 = torch.randn(Co, C, 1, 1, generator=g) / math.sqrt(C)
    b2 = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x, w1, b1, padding=1))
    if two:
        ref = ref + F.conv2d(x2, w2, b2)
    xa, xa2 = to_nhwc(x), to_nhwc(x2)
    wt1, cp1 = pack_conv(w1.to(DEV), xa.ld)
    wt2, cp2 = pack_conv(w2.to(DEV), xa2.ld)
    cw1, cw2 = ConvW(wt1, b1.to(DEV), Co, 3, 3, cp1), ConvW(wt2, b2.to(DEV), Co, 1, 1, cp2)
    xpl, x2pl = eng.to_planes(xa, "t.xpl"), eng.to_planes(xa2, "t.x2pl")
    tile = N.CONV_TILE_HI | 6
    outs, pls = [], []
    for k in (1, sk, sk):
        out = eng.new_act(f"t.sk{len(outs)}", B, H, W, Co)
        out.t.fill_(3.0)
        opl = eng.buf(f"t.skpl{len(outs)}", B * H * W * Co).view(torch.int16)
        kw = dict(in2=xa2, w2=cw2, in2_pl=x2pl) if two else {}
        eng.conv(xa, cw1, out, pad=1, act=N.ACT_RELU, tile_hint=tile, arith=2, in_pl=xpl, out_pl=opl, split_k=k, **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(out.t).all()
        assert rel_err(from_nhwc(out), ref) < 2e-5
        outs.append(out.t.clone())
        pls.append(opl.clone())
    assert torch.equal(outs[1], outs[2]) and torch.equal(pls[1], pls[2])
    assert (outs[0] - outs[1]).abs().max() < 1e-5
    want = eng.to_planes(Act(outs[1], B, H, W, Co, rup(Co, 4)), "t.wpl")
    assert torch.equal(pls[1][: want.numel()], want)


@pytest.mark.parametrize("kind,cfg", [("pl", (2, 96, 16, 16, 192, True, 3)), ("pl", (1, 384, 16, 32, 384, False, 8)), ("pl", (1, 384, 16, 16, 384, True, 12)),
                                      ("pc", (3, 8, 8, 768, 200, 1, True, True, 1, 4)), ("pc", (5, 8, 8, 64, 40, 3, False, False, 1, 2)),
                                      ("pc", (2, 15, 15, 192, 96, 2, True, True, 1, 3)), ("pc", (1, 8, 8, 3072, 768, 0, True, True, 2, 8))])
def test_k_slice_epilogue_forms_are_bit_identical(kind, cfg):
    """round 6: splitk_epilogue_vec_kernel issues every load of an item -- all K slices, bias, the slice of the second phase, residual -- before the
    first use (one round trip; the first kernel waited behind each of them in turn: 9 - 13 us for a few MB) and adds in the same order: fp32 output and
    operand planes identical bit for bit to the first form (vs_debug_set key 8 = 1), for compile-time slice counts (2 / 3 / 4 / 6 / 8) and the run-time one."""
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(77)
    res_by_form = []
    for form in (1, 0):
        N.check(eng.lib.vs_debug_set(8, form), "vs_debug_set")
        try:
            if kind == "pl":
                B, C, H, W, Co, two, sk = cfg
                gg = torch.Generator().manual_seed(31)
                x, x2 = torch.randn(B, C, H, W, generator=gg), torch.randn(B, C, H, W, generator=gg)
                w1 = torch.randn(Co, C, 3, 3, generator=gg) / math.sqrt(C * 9)
                w2 = torch.randn(Co, C, 1, 1, generator=gg) / math.sqrt(C)
                b1, b2 = torch.randn(Co, generator=gg), torch.randn(Co, generator=gg)
                xa, xa2 = to_nhwc(x), to_nhwc(x2)
                wt1, cp1 = pack_conv(w1.to(DEV), xa.ld)
                wt2, cp2 = pack_conv(w2.to(DEV), xa2.ld)
                cw1, cw2 = ConvW(wt1, b1.to(DEV), Co, 3, 3, cp1), ConvW(wt2, b2.to(DEV), Co, 1, 1, cp2)
                xpl, x2pl = eng.to_planes(xa, "t.xpl"), eng.to_planes(xa2, "t.x2pl")
                out = eng.new_act(f"t.ef{form}", B, H, W, Co)
                out.t.fill_(3.0)
                opl = eng.buf(f"t.efpl{form}", B * H * W * Co).view(torch.int16)
                kw = dict(in2=xa2, w2=cw2, in2_pl=x2pl) if two else {}
                eng.conv(xa, cw1, out, pad=1, act=N.ACT_RELU, tile_hint=N.CONV_TILE_HI | 6, arith=2, in_pl=xpl, out_pl=opl, split_k=sk, **kw)
                torch.cuda.synchronize()
                res_by_form.append((out.t.clone(), opl.clone()))
            else:
                B, H, W, K, Nn, act, grn, use_res, tl, sk = cfg
                gg = torch.Generator().manual_seed(5 + K)
                h = torch.randn(B, H * W, K, generator=gg)
                sc, sh = 1 + 0.3 * torch.randn(B, K, generator=gg), 0.1 * torch.randn(K, generator=gg)
                w = torch.randn(Nn, K, generator=gg) / math.sqrt(K)
                bias, res = torch.randn(Nn, generator=gg), torch.randn(B, H * W, Nn, generator=gg)
                xa = Act(dv(h), B, H, W, K, K)
                wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
                cw = ConvW(wt, dv(bias), Nn, 1, 1, cp)
                ra = Act(dv(res), B, H, W, Nn, Nn)
                out = Act(torch.full((B * H * W * Nn,), float("nan"), device=DEV), B, H, W, Nn, Nn)
                kw = dict(a_scale=dv(sc), a_scale_ld=K, a_shift=dv(sh)) if grn else {}
                eng.conv(xa, cw, out, act=act, res=(ra if use_res else None), tile_hint=N.CONV_TILE_HI | tl, split_k=sk, arith=2, **kw)
                torch.cuda.synchronize()
                assert torch.isfinite(out.t).all()
                res_by_form.append((out.t.clone(),))
        finally:
            N.check(eng.lib.vs_debug_set(8, 0), "vs_debug_set")
    for a_, b_ in zip(*res_by_form):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("cx", [16, 1])
def test_conv3x3_small_two_phase(eng, cx):
    """thin-layer kernel with the fused 1x1 res_conv (unet.py:38-39) + residual at a channel offset; bit-identical to the patch kernel."""
    if not eng.use_split:
        pytest.skip("split back-end only")
    g = torch.Generator().manual_seed(23)
    B, Cm, H, W, Co = 3, 16, 24, 40, 16
    t = torch.randn(B, Cm, H, W, generator=g)
    x = torch.randn(B, cx, H, W, generator=g)
    w1 = torch.randn(Co, Cm, 3, 3, generator=g) / math.sqrt(Cm * 9)
    b1 = torch.randn(Co, generator=g)
    w2 = torch.randn(Co, cx, 1, 1, generator=g) / math.sqrt(cx)
    b2 = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(t, w1, b1, padding=1)) + F.conv2d(x, w2, b2)
    ta, xa = to_nhwc(t), to_nhwc(x)
    wt1, cp1 = pack_conv(w1.to(DEV), ta.ld)
    wt2, cp2 = pack_conv(w2.to(DEV), xa.ld)
    outs = []
    for tile in (0x44, 10):
        out = eng.new_act(f"thin{tile}", B, H, W, 24)
        out.t.fill_(7.0)
        eng.conv(ta, ConvW(wt1, b1.to(DEV), Co, 3, 3, cp1), out, pad=1, act=N.ACT_RELU, in2=xa, w2=ConvW(wt2, b2.to(DEV), Co, 1, 1, cp2),
                 out_coff=4, n_store=Co, tile_hint=tile)
        torch.cuda.synchronize()
        full = out.t.view(B, H, W, 24).cpu()
        assert rel_err(full[..., 4:20].permute(0, 3, 1, 2), ref) < 2e-5
        assert (full[..., :4] == 7.0).all() and (full[..., 20:] == 7.0).all()
        outs.append(full)
    assert torch.equal(outs[0], outs[1])


class Eng(HipEngine):
    """engine without a model: only the kernel wrappers + workspace."""

    def __init__(self, use_split=True, arith=3):
        self.arith = arith
        self.dev = torch.device(DEV)
        self.lib = N.lib()
        self._ws = {}
        self.kernel_timers = None
        self.use_split = use_split
        self.autotune = False
        self.time_all_convs = False
        self._tile_cache = {}
        self.msg_table_conv = True
        self._ws_used = {}
        self.layer_arith = {}
        self.planes_chain = self.planes_splitk = self.msg0_planes = True
        self.planes_chain_ran = False
        self.grn_fold = self.grn_straddle = True
        self._calib = None


@pytest.fixture(scope="module", params=["split", "h2", "f32"])
def eng(request):
    """the arithmetic back-ends of vs_conv_gemm: 3 x bf16 split (exact), 2 x f16 split (3 products) and the fp32-input MFMA"""
    return Eng(use_split=(request.param != "f32"), arith=2 if request.param == "h2" else 3)


_KEEP = []


def dv(t):
    """device copy that stays referenced until the test module is torn down (the C-ABI only sees raw pointers,
    so a temporary freed by Python could be recycled by the caching allocator before the kernel runs)."""
    t = t.to(DEV).contiguous()
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def rel_err(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, pad_mode, act, tile
    (2, 16, 17, 23, 16, 3, 1, 1, 0, 1, 0),
    (1, 1, 32, 32, 16, 3, 1, 1, 0, 1, 0),
    (3, 32, 20, 20, 64, 3, 2, 1, 0, 0, 0),
    (2, 64, 16, 16, 136, 3, 1, 1, 1, 0, 0),
    (2, 384, 8, 8, 384, 3, 1, 1, 0, 1, 1),
    (1, 48, 9, 11, 40, 1, 1, 0, 0, 2, 2),
    (2, 24, 12, 12, 20, 3, 1, 1, 1, 3, 3),
    (2, 64, 33, 31, 96, 3, 1, 1, 0, 1, 2),
    # producer/consumer kernels (tile codes 6..9), incl. ragged M / N, reflect, stride 2, tiny K
    (2, 384, 16, 16, 384, 3, 1, 1, 0, 1, 1),
    (3, 64, 19, 23, 136, 3, 1, 1, 1, 2, 1),
    (2, 32, 20, 20, 64, 3, 2, 1, 0, 0, 2),
    (1, 16, 40, 24, 40, 1, 1, 0, 0, 3, 2),
    (2, 1, 32, 32, 16, 3, 1, 1, 0, 1, 2),
    (2, 96, 9, 9, 200, 3, 1, 1, 0, 1, 1),
    (2, 128, 16, 16, 96, 1, 1, 0, 0, 2, 5),
    (2, 128, 16, 16, 192, 3, 1, 1, 0, 1, 4),
    # 3x3 patch kernel (tile codes 10..12): aligned and ragged frames, zero / reflect padding, small and large channel counts
    (2, 16, 32, 32, 16, 3, 1, 1, 0, 1, 10),
    (2, 384, 16, 16, 384, 3, 1, 1, 0, 1, 12),
    (3, 64, 24, 48, 40, 3, 1, 1, 1, 2, 11),
    (2, 32, 13, 21, 136, 3, 1, 1, 1, 0, 12),
    (1, 1, 32, 32, 16, 3, 1, 1, 0, 1, 10),
    (2, 100, 9, 17, 64, 3, 1, 1, 0, 3, 11),
    (2, 48, 16, 16, 48, 3, 1, 1, 0, 1, 0),
    (2, 96, 8, 8, 200, 1, 1, 0, 0, 2, 13),
    # wave-specialised patch kernel (tile 15): aligned / ragged frames, ragged N, reflect, single chunk, odd chunk counts
    (2, 384, 16, 16, 384, 3, 1, 1, 0, 1, 15),
    (3, 48, 13, 21, 136, 3, 1, 1, 1, 2, 15),
    (2, 16, 32, 32, 16, 3, 1, 1, 0, 1, 15),
    (1, 1, 8, 16, 130, 3, 1, 1, 0, 3, 15),
    (2, 100, 24, 48, 300, 3, 1, 1, 0, 0, 15),
    (2, 384, 16, 16, 384, 3, 1, 1, 0, 1, 0x40),     # 0x40 = tile 16: 128 x 192
    (3, 48, 13, 21, 200, 3, 1, 1, 1, 2, 0x40),
    (1, 16, 8, 16, 16, 3, 1, 1, 0, 3, 0x40),
    # persistent thin-layer kernel (tile 20 = 0x44): 16 input channels, <= 32 outputs; ragged frames, reflect, 1-channel input
    (2, 16, 32, 48, 16, 3, 1, 1, 0, 1, 0x44),
    (3, 16, 13, 21, 32, 3, 1, 1, 1, 2, 0x44),
    (2, 1, 24, 32, 16, 3, 1, 1, 0, 1, 0x44),
    (5, 12, 40, 40, 20, 3, 1, 1, 0, 3, 0x44),
    (2, 768, 16, 16, 64, 3, 1, 1, 0, 1, 0x43),      # 0x43 = tile 19: 128 x 64
    (2, 768, 32, 16, 64, 3, 1, 1, 0, 1, 0x45),      # 0x45 = tile 21: 256 px x 64 ch
    (3, 80, 21, 37, 50, 3, 1, 1, 1, 2, 0x45),
    (3, 48, 13, 21, 70, 3, 1, 1, 1, 2, 0x43),
    (3, 160, 7, 9, 130, 3, 2, 1, 0, 1, 14),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_gemm_matches_conv2d(eng, case):
    B, Cin, H, W, Cout, k, s, p, pm, act, tile = case
    if tile >= 10 and not eng.use_split:
        pytest.skip("patch / wave-specialised kernels exist for the split back-end only")
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    xp = F.pad(x, (p, p, p, p), mode="reflect") if (pm and p) else x
    ref = F.conv2d(xp, w, b, stride=s, padding=0 if pm else p)
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref), 3: torch.tanh(ref)}[act]
    xa = to_nhwc(x)
    wt, cp = pack_conv(w.to(DEV), xa.ld)
    out = eng.new_act("o", B, ref.shape[2], ref.shape[3], Cout)
    out.t.fill_(float("nan"))
    eng.conv(xa, ConvW(wt, b.to(DEV), Cout, k, k, cp), out, stride=s, pad=p, pad_mode=pm, act=act, tile_hint=tile)
    torch.cuda.synchronize()
    got = from_nhwc(out)
    assert rel_err(got, ref) < 2e-5          # fp32 accumulate, different summation order only
    full = out.t.view(B, out.H, out.W, out.ld).cpu()
    assert torch.isfinite(full).all() and (full[..., Cout:] == 0).all()   # pad lanes written as zero


@pytest.mark.parametrize("tile", [0, 2, 10, 11, 1, 15, 0x40, 0x43, 0x45])
def test_conv_two_phase_residual_block(eng, tile):
    """ResnetBlock tail: relu(conv3x3(t)+b) + (conv1x1(x)+b2), written at a channel offset of a wider buffer."""
    if tile >= 10 and not eng.use_split:
        pytest.skip("split back-end only")
    g = torch.Generator().manual_seed(5)
    B, Cm, Cx, H, W, Co = 2, 32, 16, 14, 18, 32
    t = torch.randn(B, Cm, H, W, generator=g)
    x = torch.randn(B, Cx, H, W, generator=g)
    w1 = torch.randn(Co, Cm, 3, 3, generator=g) / math.sqrt(Cm * 9)
    b1 = torch.randn(Co, generator=g)
    w2 = torch.randn(Co, Cx, 1, 1, generator=g) / math.sqrt(Cx)
    b2 = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(t, w1, b1, padding=1)) + F.conv2d(x, w2, b2)
    ta, xa = to_nhwc(t), to_nhwc(x)
    wt1, cp1 = pack_conv(w1.to(DEV), ta.ld)
    wt2, cp2 = pack_conv(w2.to(DEV), xa.ld)
    out = eng.new_act("wide", B, H, W, 48)
    out.t.fill_(7.0)
    eng.conv(ta, ConvW(wt1, b1.to(DEV), Co, 3, 3, cp1), out, pad=1, act=N.ACT_RELU, in2=xa, w2=ConvW(wt2, b2.to(DEV), Co, 1, 1, cp2),
             out_coff=8, n_store=Co, tile_hint=tile)
    torch.cuda.synchronize()
    full = out.t.view(B, H, W, 48).cpu()
    assert rel_err(full[..., 8:40].permute(0, 3, 1, 2), ref) < 2e-5
    assert (full[..., :8] == 7.0).all() and (full[..., 40:] == 7.0).all()   # neighbours untouched


@pytest.mark.parametrize("tile", [0, 2])
def test_conv_grn_transform_and_residual(eng, tile):
    """pwconv2 with the GRN apply folded into the A load and the block residual (convnext.py:50-56)."""
    if tile >= 10 and not eng.use_split:
        pytest.skip("split back-end only")
    g = torch.Generator().manual_seed(6)
    B, HW, K, Nn = 3, 50, 72, 20
    h = torch.randn(B, HW, K, generator=g)
    sc = 1 + 0.3 * torch.randn(B, K, generator=g)
    sh = 0.1 * torch.randn(K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    res = torch.randn(B, HW, Nn, generator=g)
    ref = res + F.linear(h * sc[:, None, :] + sh, w, bias)
    ha = Act(h.to(DEV).contiguous(), B, HW, 1, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    ra = Act(res.to(DEV).contiguous(), B, HW, 1, Nn, Nn)
    shp = torch.zeros(cp); shp[:K] = sh
    eng.conv(ha, ConvW(wt, bias.to(DEV), Nn, 1, 1, cp), ra, res=ra, a_scale=sc.to(DEV).contiguous(), a_scale_ld=K, a_shift=shp.to(DEV), tile_hint=tile)
    torch.cuda.synchronize()
    assert rel_err(ra.t.cpu().view(B, HW, Nn), ref) < 2e-5


@pytest.mark.parametrize("tile", [0, 1, 3, 5, 13, 0x41, 0x42])
def test_conv_epilogue_grn_partials(eng, tile):
    """pwconv1 + GELU with the GRN sum-of-squares partials written by the epilogue, then vs_grn_scale_from_partials == common.py:166-168."""
    if tile >= 10 and not eng.use_split:
        pytest.skip("split back-end only")
    g = torch.Generator().manual_seed(9)
    B, H, W, K, Nn = 3, 8, 12, 64, 200         # HW = 96 = 3 groups of 32 rows per frame
    x = torch.randn(B, H * W, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    gamma = torch.randn(Nn, generator=g)
    h = F.gelu(F.linear(x, w, bias))
    gx = torch.linalg.vector_norm(h, dim=1, keepdim=True)                     # [B,1,N]
    ref_scale = 1 + gamma * (gx / (gx.mean(dim=-1, keepdim=True) + 1e-6))
    xa = Act(x.to(DEV).contiguous(), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    out = eng.new_act("gh", B, H, W, Nn)
    guard = torch.full((B * (H * W // 32) * Nn + 4096,), float("nan"), device=DEV)      # rows 288 = 2.25 tiles of 128: ragged last tile
    part = guard[:B * (H * W // 32) * Nn]
    eng.conv(xa, ConvW(wt, bias.to(DEV), Nn, 1, 1, cp), out, act=N.ACT_GELU, tile_hint=tile, sumsq=part)
    scale = torch.empty(B, out.ld, device=DEV)
    N.check(eng.lib.vs_grn_scale_from_partials(N.ptr(part), B, H * W, Nn, N.ptr(gamma.to(DEV)), N.ptr(scale), out.ld, N.stream()), "grn")
    torch.cuda.synchronize()
    assert rel_err(out.t.view(B, H * W, out.ld)[..., :Nn].cpu(), h) < 2e-5
    assert rel_err(scale.cpu()[:, :Nn], ref_scale[:, 0]) < 2e-5
    assert torch.isnan(guard[part.numel():]).all()          # nothing written past the [M/32][N] partials


GEMM_PL_CASES = [
    # B, H, W, K, N, act, grn, res, tile (24 / 25), split_k, sumsq
    (2, 16, 16, 96, 384, 2, False, False, 24, 1, True),      # pwconv1-like: GELU + GRN partials
    (2, 16, 16, 384, 96, 0, True, True, 25, 1, False),       # pwconv2-like: GRN apply in the planes conversion + residual
    (3, 8, 8, 768, 200, 0, True, True, 24, 1, False),        # ragged M (192 rows) and N
    (3, 8, 8, 768, 200, 1, True, True, 24, 4, False),        # K split in 4 slices + epilogue kernel
    (1, 8, 8, 3072, 768, 0, True, True, 24, 8, False),
    (5, 8, 8, 64, 40, 3, False, False, 25, 2, False),        # two K16 steps per slice, tanh
    (2, 16, 24, 16, 130, 2, False, True, 24, 1, False),      # a single K step
    (3, 15, 15, 160, 96, 0, True, True, 25, 1, False),       # 225-row frames (ChunkySeal)
    (2, 31, 31, 96, 200, 2, False, False, 24, 1, True),      # 1922 rows: ragged last tile with GRN partials
    # >= 4 x 256 tiles: the XCD-aware grouped tile order of round 5 (ChunkySeal's launches); 60 x 18 tiles, both edges ragged, 1080 % 8 != 0
    (15, 31, 33, 48, 3272, 2, False, False, 24, 1, True),
    (16, 32, 32, 64, 3072, 0, True, True, 24, 1, False),     # 64 x 16 tiles, residual
    (10, 40, 40, 32, 2100, 0, False, False, 25, 1, False),   # 128-column tiles (tile 25): 63 x 17 = 1071 tiles, grouped order with a short last group
]


@pytest.mark.parametrize("case", GEMM_PL_CASES)
def test_gemm_planes_kernel(case):
    """all-DMA 1x1 GEMM on pre-split operand planes (tile codes 24 / 25, gemm_pl.hip) against torch, and bit-identical to the generic
    kernel in the same 2 x f16 arithmetic when K is not split (same products, same K order).  The GRN apply runs in the planes
    conversion (vs_to_planes_affine) instead of on the A load."""
    B, H, W, K, Nn, act, grn, res, tile, sk, sumsq = case
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(31)
    HW = H * W
    x = torch.randn(B, HW, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    scale = 1 + 0.5 * torch.randn(B, K, generator=g)
    shift = torch.randn(K, generator=g)
    r = torch.randn(B, HW, Nn, generator=g)
    a = x * scale[:, None, :] + shift if grn else x
    ref = F.linear(a, w, bias)
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref), 3: torch.tanh(ref)}[act]
    if res:
        ref = ref + r
    xa = Act(x.to(DEV).contiguous(), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    cw = ConvW(wt, bias.to(DEV), Nn, 1, 1, cp)
    ra = Act(r.to(DEV).contiguous(), B, H, W, Nn, Nn)
    pl = eng.buf("t.gpl", B * HW * K).view(torch.int16)
    am = 1.0 if grn else 16.0          # engine.A_MUL_GRN / A_MUL: the range scale the fused GRN path of the other kernels uses
    N.check(eng.lib.vs_to_planes_affine(N.ptr(xa.t), xa.rows, K, xa.ld, am, N.ptr(dv(scale)) if grn else None, K,
                                        N.ptr(dv(shift)) if grn else None, HW, N.ptr(pl), N.stream()), "to_planes_affine")
    outs = []
    for t in (N.CONV_TILE_HI | (tile - 16), 1):
        out = eng.new_act(f"t.gplo{t}", B, H, W, Nn)
        out.t.fill_(5.0)
        part = torch.full((((B * HW + 31) // 32) * Nn,), float("nan"), device=DEV) if sumsq else None
        kw = dict(act=act, res=(ra if res else None), tile_hint=t, arith=2)
        if t != 1:
            kw.update(in_pl=pl, split_k=sk, sumsq=part, a_mul=am)
        else:
            kw.update(a_scale=(dv(scale) if grn else None), a_scale_ld=K, a_shift=(dv(shift) if grn else None), sumsq=part)
        eng.conv(xa, cw, out, **kw)
        torch.cuda.synchronize()
        got = out.t.view(B, HW, out.ld)
        assert rel_err(got[..., :Nn].cpu(), ref) < 2e-5
        assert (got[..., Nn:] == 0).all()
        outs.append((out.t.clone(), part))
    if sk == 1:
        assert torch.equal(outs[0][0], outs[1][0])
        if sumsq:
            assert torch.equal(outs[0][1], outs[1][1])


GEMM_PC_CASES = [
    # B, H, W, K, N, act, grn, res, tile, split_k
    (2, 16, 16, 96, 384, 2, False, False, 1, 1),      # pwconv1-like: GELU epilogue
    (2, 16, 16, 384, 96, 0, True, True, 1, 1),        # pwconv2-like: GRN transform on the A load + residual
    (3, 8, 8, 768, 200, 0, True, True, 2, 1),         # 2 frames per 128-row tile, ragged M (192 rows) and N, 192-wide tile
    (3, 8, 8, 768, 200, 1, True, True, 1, 4),         # K split in 4 slices + epilogue kernel
    (1, 8, 8, 3072, 768, 0, True, True, 2, 8),
    (5, 8, 8, 64, 40, 3, False, False, 1, 2),         # one K pair per slice, tanh
    (2, 16, 24, 32, 130, 2, False, True, 2, 1),       # single pair, three N tiles of which one ragged
    (3, 15, 15, 160, 96, 0, True, True, 1, 1),        # 225-row frames: tiles straddle frame boundaries at arbitrary rows (ChunkySeal)
    (2, 31, 31, 96, 200, 0, True, True, 2, 1),
    # tile code 26 (round 5): 128 x 96 tiles, four consumer waves stacked over the rows (2 x f16 arithmetic only)
    (2, 16, 16, 1536, 384, 0, True, True, 10, 1),     # ConvNeXt stage-2 pwconv2: the whole K in one workgroup, no K slices
    (2, 16, 16, 96, 384, 2, False, False, 10, 1),     # GELU epilogue
    (3, 8, 8, 768, 200, 0, True, True, 10, 1),        # ragged M (192 rows) and N (200 = 2 x 96 + 8), two frames per tile
    (5, 8, 8, 64, 40, 3, False, False, 10, 2),        # K slices + tanh
    (3, 15, 15, 160, 96, 0, True, True, 10, 1),       # frame boundaries at arbitrary rows
    (2, 31, 31, 96, 200, 1, True, True, 10, 1),
]


@pytest.mark.parametrize("case", GEMM_PC_CASES)
def test_gemm1x1_pc(eng, case):
    """wave-specialised 1x1 GEMM (tile codes 17 / 18) vs torch and vs the generic kernel (bit-identical when K is not split)."""
    if not eng.use_split:
        pytest.skip("split back-end only")
    B, H, W, K, Nn, act, grn, use_res, tl, sk = case
    if tl == 10 and eng.arith != 2:
        pytest.skip("tile 26 exists in the 2 x f16 arithmetic only")
    g = torch.Generator().manual_seed(hash(case) % 1000)
    h = torch.randn(B, H * W, K, generator=g)
    sc = 1 + 0.3 * torch.randn(B, K, generator=g)
    sh = 0.1 * torch.randn(K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    res = torch.randn(B, H * W, Nn, generator=g)
    a = h * sc[:, None, :] + sh if grn else h
    ref = F.linear(a, w, bias)
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref), 3: torch.tanh(ref)}[act]
    if use_res:
        ref = ref + res
    ha = Act(h.to(DEV).contiguous(), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    cw = ConvW(wt, bias.to(DEV), Nn, 1, 1, cp)
    ld = (Nn + 3) // 4 * 4
    kw = dict(act=act)
    if grn:
        kw.update(a_scale=sc.to(DEV).contiguous(), a_scale_ld=K, a_shift=sh.to(DEV).contiguous())
    outs = []
    for hint, k in ((N.CONV_TILE_HI | tl, sk), (1, 1)):
        ra = Act(torch.full((B * H * W * ld,), float("nan"), device=DEV), B, H, W, Nn, ld)
        rr = None
        if use_res:
            rr = Act(torch.zeros(B * H * W * ld, device=DEV), B, H, W, Nn, ld)
            rr.t.view(B, H * W, ld)[..., :Nn] = res.to(DEV)
        eng.conv(ha, cw, ra, res=rr, tile_hint=hint, split_k=k, **kw)
        torch.cuda.synchronize()
        full = ra.t.view(B, H * W, ld).cpu()
        assert rel_err(full[..., :Nn], ref) < 2e-5
        assert (full[..., Nn:] == 0).all()
        outs.append(full)
    if sk == 1:
        assert torch.equal(outs[0], outs[1])      # same arithmetic, same K order as the generic kernel


@pytest.mark.parametrize("stride", [4, 2])
def test_patch_conv_stem(eng, stride):
    """4x4 stem (stride 4, and ChunkySeal's stride 2) as a 4x1 conv over 16-float pixel runs."""
    g = torch.Generator().manual_seed(7)
    B, S, Co = 2, 36, 24
    x = torch.randn(B, 3, S, S, generator=g)
    w = torch.randn(Co, 3, 4, 4, generator=g) / 7
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, stride=stride)
    xa = to_nhwc(x, 4)
    wt, cp = pack_patch_conv(w.to(DEV), 4)
    Ho = (S - 4) // stride + 1
    out = eng.new_act("stem", B, Ho, Ho, Co)
    eng.conv(xa, ConvW(wt, b.to(DEV), Co, 4, 1, cp), out, geom=(Ho, stride * 4, 16, stride, 1, 0, 0))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out), ref) < 2e-5


@pytest.mark.parametrize("hw", [(16, 16), (15, 13)])
def test_patch_conv_downsample(eng, hw):
    g = torch.Generator().manual_seed(8)
    B, Cc, Co = 2, 24, 40
    H, W = hw
    x = torch.randn(B, Cc, H, W, generator=g)
    w = torch.randn(Co, Cc, 2, 2, generator=g) / 10
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, stride=2)
    xa = to_nhwc(x)
    wt, cp = pack_patch_conv(w.to(DEV), xa.ld)
    out = eng.new_act("dn", B, H // 2, W // 2, Co)
    eng.conv(xa, ConvW(wt, b.to(DEV), Co, 2, 1, cp), out, geom=(W // 2, 2 * xa.ld, 2 * xa.ld, 2, 1, 0, 0))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out), ref) < 2e-5


@pytest.mark.parametrize("C_,act", [(16, 1), (96, 0), (362, 2), (768, 2)])
def test_layernorm_act(eng, C_, act):
    g = torch.Generator().manual_seed(C_)
    x = torch.randn(2, C_, 9, 7, generator=g) * 3 + 1
    w, b = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g)
    u = x.mean(1, keepdim=True); s = (x - u).pow(2).mean(1, keepdim=True)
    ref = w[:, None, None] * ((x - u) / torch.sqrt(s + 1e-6)) + b[:, None, None]
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref)}[act]
    xa = to_nhwc(x)
    out = eng.new_act("ln", 2, 9, 7, C_)
    eng.layernorm(xa, w.to(DEV), b.to(DEV), out, act=act)
    torch.cuda.synchronize()
    assert (from_nhwc(out) - ref).abs().max() < 2e-5


@pytest.mark.parametrize("C_,H,W", [(96, 16, 16), (24, 9, 13), (362, 7, 7), (768, 8, 8), (192, 40, 36), (96, 33, 19), (20, 64, 64), (130, 17, 50)])
def test_dwconv7_ln(eng, C_, H, W):
    g = torch.Generator().manual_seed(C_ + H)
    B = 2
    x = torch.randn(B, C_, H, W, generator=g)
    wd = torch.randn(C_, 1, 7, 7, generator=g) / 7
    bd = torch.randn(C_, generator=g) * 0.1
    lw, lb = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.1
    ref = F.conv2d(x, wd, bd, padding=3, groups=C_).permute(0, 2, 3, 1)
    ref = F.layer_norm(ref, (C_,), lw, lb, 1e-6).permute(0, 3, 1, 2)
    xa = to_nhwc(x)
    ld = xa.ld
    wp = torch.zeros(49, ld); wp[:, :C_] = wd.reshape(C_, 49).t()
    pad = lambda v: dv(torch.cat([v, torch.zeros(ld - C_)]))   # noqa: E731
    out = eng.new_act("dw", B, H, W, C_)
    N.check(eng.lib.vs_dwconv7_ln(N.ptr(xa.t), B, H, W, C_, ld, N.ptr(dv(wp)), N.ptr(pad(bd)), N.ptr(pad(lw)), N.ptr(pad(lb)), 1e-6,
                                  N.ptr(out.t), out.ld, N.stream()), "dw")
    torch.cuda.synchronize()
    assert (from_nhwc(out) - ref).abs().max() < 3e-5
    # planes output (the operand of the all-DMA pwconv1 GEMM) == vs_to_planes of the fp32 output, padded to the consumer's K
    Cp = rup(C_, 32)
    pl = torch.full((2 * B * H * W * Cp,), 77, dtype=torch.int16, device=DEV)
    N.check(eng.lib.vs_dwconv7_ln_planes(N.ptr(xa.t), B, H, W, C_, ld, N.ptr(dv(wp)), N.ptr(pad(bd)), N.ptr(pad(lw)), N.ptr(pad(lb)), 1e-6,
                                         16.0, Cp, N.ptr(pl), N.stream()), "dw planes")
    wide = torch.zeros(B * H * W, Cp, device=DEV)
    wide[:, :C_] = out.t.view(B * H * W, out.ld)[:, :C_]
    want = torch.empty_like(pl)
    N.check(eng.lib.vs_to_planes(N.ptr(wide), B * H * W, Cp, Cp, 16.0, N.ptr(want), N.stream()), "to_planes")
    torch.cuda.synchronize()
    assert torch.equal(pl, want)


@pytest.mark.parametrize("C_,HW", [(384, 4096), (100, 37), (1448, 225)])
def test_grn_scale(eng, C_, HW):
    g = torch.Generator().manual_seed(C_)
    B = 3
    h = torch.randn(B, HW, C_, generator=g)
    gamma = torch.randn(C_, generator=g)
    gx = torch.norm(h, p=2, dim=1, keepdim=True)
    ref = 1 + gamma * (gx / (gx.mean(dim=-1, keepdim=True) + 1e-6))
    ld = rup(C_, 4)
    hp = torch.zeros(B, HW, ld); hp[..., :C_] = h
    part = torch.empty(((HW + 63) // 64) * B * C_, device=DEV)
    scale = torch.full((B, ld), float("nan"), device=DEV)
    N.check(eng.lib.vs_grn_scale(N.ptr(dv(hp)), B, HW, C_, ld, N.ptr(dv(gamma)), N.ptr(part), N.ptr(scale), N.stream()), "grn")
    torch.cuda.synchronize()
    assert (scale.cpu()[:, :C_] - ref[:, 0]).abs().max() < 1e-5
    assert (scale.cpu()[:, C_:] == 0).all()


def test_upcat2x(eng):
    g = torch.Generator().manual_seed(9)
    B, C1, C2, H, W = 2, 24, 8, 7, 9
    x, sk = torch.randn(B, C1, H, W, generator=g), torch.randn(B, C2, H, W, generator=g)
    ref = F.interpolate(torch.cat((x, sk * 2 ** -0.5), 1), scale_factor=2, mode="bilinear", align_corners=False)
    xa, sa = to_nhwc(x), to_nhwc(sk)
    out = eng.new_act("uc", B, 2 * H, 2 * W, C1 + C2)
    N.check(eng.lib.vs_upcat2x(N.ptr(xa.t), C1, xa.ld, N.ptr(sa.t), C2, sa.ld, 2 ** -0.5, B, H, W, N.ptr(out.t), out.ld, N.stream()), "uc")
    torch.cuda.synchronize()
    assert (from_nhwc(out) - ref).abs().max() < 1e-6


@pytest.mark.parametrize("shape", [(2, 24, 8, 7, 9, 16), (1, 64, 64, 8, 16, 32), (2, 40, 24, 5, 6, 64), (1, 32, 32, 3, 3, 128),
                                   (2, 16, 16, 1, 2, 16), (2, 32, 32, 19, 13, 16), (1, 48, 16, 9, 17, 32)])
def test_upconv_lowres_gemm_gather_ln(eng, shape):
    """Upsample group (common.py:45-52) as cat2 -> 1x1 GEMM of the nine taps at the LOW resolution -> gather + LayerNorm + ReLU,
    against bilinear x2 -> ReflectionPad2d(1) -> Conv3x3 -> LayerNorm(channels_first) -> ReLU in torch fp32."""
    B, C1, C2, H, W, Co = shape
    g = torch.Generator().manual_seed(31)
    x, sk = torch.randn(B, C1, H, W, generator=g), torch.randn(B, C2, H, W, generator=g)
    w = torch.randn(Co, C1 + C2, 3, 3, generator=g) / math.sqrt(9 * (C1 + C2))
    lw, lb = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    up = F.interpolate(torch.cat((x, sk * 2 ** -0.5), 1), scale_factor=2, mode="bilinear", align_corners=False)
    cv = F.conv2d(F.pad(up, (1, 1, 1, 1), mode="reflect"), w)
    u = cv.mean(1, keepdim=True)
    sdev = (cv - u).pow(2).mean(1, keepdim=True)
    ref = F.relu(lw[None, :, None, None] * ((cv - u) / torch.sqrt(sdev + 1e-6)) + lb[None, :, None, None])
    assert eng.lib.vs_upconv_supported(Co) == 1
    xa, sa = to_nhwc(x), to_nhwc(sk)
    lc = eng.new_act("t.lcat", B, H, W, C1 + C2)
    N.check(eng.lib.vs_cat2_scale(N.ptr(xa.t), C1, xa.ld, N.ptr(sa.t), C2, sa.ld, 2 ** -0.5, lc.rows, N.ptr(lc.t), lc.ld, N.stream()), "cat2")
    wz, cpz = pack_conv(w.to(DEV).permute(2, 3, 0, 1).reshape(9 * Co, C1 + C2)[:, :, None, None], lc.ld)
    z = eng.new_act("t.z", B, H, W, 9 * Co)
    eng.conv(lc, ConvW(wz, None, 9 * Co, 1, 1, cpz), z)
    out = eng.new_act("t.upln", B, 2 * H, 2 * W, Co)
    N.check(eng.lib.vs_upconv_gather_ln(N.ptr(z.t), z.ld, B, H, W, Co, N.ptr(dv(lw)), N.ptr(dv(lb)), 1e-6, N.ACT_RELU, N.ptr(out.t),
                                        out.ld, N.stream()), "upconv_gather_ln")
    torch.cuda.synchronize()
    assert (from_nhwc(out) - ref).abs().max() < 2e-5          # LayerNorm output is O(1); fp32 re-association only
    if eng.use_split and eng.lib.vs_upconv_fused_supported(C1, C2, Co):      # one-kernel form (z stays in LDS), same maths
        out2 = eng.new_act("t.upln2", B, 2 * H, 2 * W, Co)
        out2.t.fill_(-3.0)
        cw = ConvW(wz, None, 9 * Co, 1, 1, cpz).with_split(eng.arith)
        N.check(eng.lib.vs_upconv_fused(N.ptr(xa.t), C1, xa.ld, N.ptr(sa.t), C2, sa.ld, 2 ** -0.5, N.ptr(cw.split), B, H, W, Co,
                                        N.ptr(dv(lw)), N.ptr(dv(lb)), 1e-6, N.ACT_RELU, N.ptr(out2.t), out2.ld, eng.arith, 16.0,
                                        1.0 / (16.0 * cw.w_mul), N.stream()), "upconv_fused")
        torch.cuda.synchronize()
        assert (from_nhwc(out2) - ref).abs().max() < 2e-5
    # x == NULL: the producer already wrote columns [0, C1)
    lc.t.view(B, H, W, lc.ld)[..., C1:] = -5.0
    N.check(eng.lib.vs_cat2_scale(None, C1, 0, N.ptr(sa.t), C2, sa.ld, 2 ** -0.5, lc.rows, N.ptr(lc.t), lc.ld, N.stream()), "cat2")
    torch.cuda.synchronize()
    got = lc.t.view(B, H, W, lc.ld).cpu()
    assert torch.equal(got[..., :C1], x.permute(0, 2, 3, 1)) and torch.equal(got[..., C1:], (sk * 2 ** -0.5).permute(0, 2, 3, 1))


@pytest.mark.parametrize("shape", [(37, 16), (130, 64), (9, 320)])
def test_rmsnorm_act(eng, shape):
    """ChanRMSNorm (common.py:172-179) + SiLU (+ residual branch) against F.normalize * sqrt(C) * gamma in torch fp32"""
    rows, Cc = shape
    g = torch.Generator().manual_seed(41)
    x, add = torch.randn(rows, Cc, generator=g), torch.randn(rows, Cc, generator=g)
    gamma = torch.rand(Cc, generator=g) + 0.5
    ref = F.silu(F.normalize(x, dim=1) * Cc ** 0.5 * gamma) + add
    out = torch.full((rows, Cc + 4), -7.0, device=DEV)
    N.check(eng.lib.vs_rmsnorm_act(N.ptr(dv(x)), rows, Cc, Cc, N.ptr(dv(gamma)), N.ACT_SILU, N.ptr(dv(add)), Cc, N.ptr(out), Cc + 4, N.stream()), "rms")
    torch.cuda.synchronize()
    assert (out.cpu()[:, :Cc] - ref).abs().max() < 2e-6 and (out.cpu()[:, Cc:] == 0).all()


@pytest.mark.parametrize("cfg", [(2, 16, 16, 6, 64, 0), (2, 16, 16, 6, 64, 8), (3, 8, 8, 2, 16, 4), (1, 8, 12, 3, 32, 0), (2, 8, 8, 2, 16, 0),
                                 # matrix-core kernel: 64 / 128 / 256 tokens per group x head sizes 16 / 32 / 64 (the first two rows above as well)
                                 (1, 8, 16, 2, 32, 0), (1, 16, 16, 2, 16, 0), (2, 16, 8, 3, 32, 8), (1, 16, 16, 1, 32, 0), (1, 16, 8, 2, 64, 0),
                                 (2, 32, 16, 2, 16, 8)])
def test_vit_attention(eng, cfg):
    """vit.py:302-360 Attention.forward core (scaled q.k^T + decomposed relative positions, softmax, @v) incl. the window partition
    of vit.py:363-402, against the same maths in torch fp32"""
    B, H, W, heads, hd, win = cfg
    D = heads * hd
    g = torch.Generator().manual_seed(43)
    qkv = torch.randn(B, H, W, 3 * D, generator=g)
    Th, Tw = (win, win) if win else (H, W)
    rel_h, rel_w = 0.3 * torch.randn(2 * Th - 1, hd, generator=g), 0.3 * torch.randn(2 * Tw - 1, hd, generator=g)
    x = qkv
    if win:
        x = x.view(B, H // win, win, W // win, win, 3 * D).permute(0, 1, 3, 2, 4, 5).reshape(-1, win, win, 3 * D)
    Bw = x.shape[0]
    q, k, v = x.reshape(Bw, Th * Tw, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bw * heads, Th * Tw, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    Rh = rel_h[torch.arange(Th)[:, None] - torch.arange(Th)[None, :] + Th - 1]
    Rw = rel_w[torch.arange(Tw)[:, None] - torch.arange(Tw)[None, :] + Tw - 1]
    rq = q.reshape(Bw * heads, Th, Tw, hd)
    attn = (attn.view(Bw * heads, Th, Tw, Th, Tw) + torch.einsum("bhwc,hkc->bhwk", rq, Rh)[:, :, :, :, None]
            + torch.einsum("bhwc,wkc->bhwk", rq, Rw)[:, :, :, None, :]).view(Bw * heads, Th * Tw, Th * Tw)
    o = (attn.softmax(-1) @ v).view(Bw, heads, Th, Tw, hd).permute(0, 2, 3, 1, 4).reshape(Bw, Th, Tw, D)
    if win:
        o = o.view(B, H // win, W // win, win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, D)
    out = torch.empty(B, H, W, D, device=DEV)
    N.check(eng.lib.vs_vit_attention(N.ptr(dv(qkv)), B, H, W, heads, hd, win, N.ptr(dv(rel_h)), N.ptr(dv(rel_w)), N.ptr(out), N.stream()), "attn")
    torch.cuda.synchronize()
    assert (out.cpu() - o).abs().max() < 1e-5


@pytest.mark.parametrize("Bm", [1, 4])
def test_first_bottleneck_block_message_table(eng, Bm):
    """unet.py:183-185 first bottleneck ResnetBlock on [latent | spatially constant message]: the table form (vs_msg_pre + VS_CONV_PRE,
    K over the latent channels only) against the plain block on the materialised 384-channel map, and against torch fp32"""
    if not eng.use_split:
        pytest.skip("split back-end only")
    B, nlat, hid, H, W, Co = 4, 128, 256, 16, 32, 384
    g = torch.Generator().manual_seed(51)
    latent = torch.randn(B, nlat, H, W, generator=g)
    msg = torch.randn(Bm, hid, generator=g)
    x = torch.cat([latent, msg[:, :, None, None].expand(Bm, hid, H, W).expand(B, hid, H, W) if Bm == 1 else msg[:, :, None, None].expand(B, hid, H, W)], 1)
    w0 = torch.randn(Co, nlat + hid, 3, 3, generator=g) / math.sqrt(9 * (nlat + hid))
    w1 = torch.randn(Co, Co, 3, 3, generator=g) / math.sqrt(9 * Co)
    wr = torch.randn(Co, nlat + hid, 1, 1, generator=g) / math.sqrt(nlat + hid)
    b0, b1, br = (torch.randn(Co, generator=g) * 0.1 for _ in range(3))
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, b0, padding=1)), w1, b1, padding=1)) + F.conv2d(x, wr, br)
    xa = to_nhwc(x)
    p = {}
    for k, (w, b) in dict(c0=(w0, b0), c1=(w1, b1), res=(wr, br)).items():
        wt, cp = pack_conv(w.to(DEV), rup(w.shape[1], 4))
        p[k] = ConvW(wt, b.to(DEV), Co, w.shape[2], w.shape[3], cp)
    p["cout"] = Co
    plain = from_nhwc(eng.resblock(xa, p, "tb")).clone()
    lat_dev = dv(msg)
    out = from_nhwc(eng.resblock_msg0(xa, p, "tm", lat_dev, Bm, nlat))
    torch.cuda.synchronize()
    assert rel_err(plain, ref) < 2e-5
    assert rel_err(out, ref) < 2e-5 and (out - plain).abs().max() < 2e-5


@pytest.mark.parametrize("B", [32, 8])
def test_first_bottleneck_block_on_operand_planes(eng, B):
    """round 5: resblock_msg0(planes_out=True) -- c0 (latent planes + border-class table epilogue) and c1 (+ 1x1 phase over h3) on the all-DMA
    planes kernel, output as the chain's operand planes.  32 frames (256 tiles, no K split): the planes equal the split of the fp32 result of
    the wave-specialised path BIT FOR BIT (same products, same K order); 8 frames (K slices in c1): equal to fp32 rounding."""
    if not (eng.use_split and eng.arith == 2 and eng.planes_chain):
        pytest.skip("2 x f16 planes chain only")
    nlat, hid, H, W, Co = 128, 256, 32, 32, 384
    g = torch.Generator().manual_seed(52)
    latent = torch.randn(B, nlat, H, W, generator=g)
    msg = torch.randn(1, hid, generator=g)
    x = torch.cat([latent, msg[:, :, None, None].expand(B, hid, H, W)], 1)
    w0 = torch.randn(Co, nlat + hid, 3, 3, generator=g) / math.sqrt(9 * (nlat + hid))
    w1 = torch.randn(Co, Co, 3, 3, generator=g) / math.sqrt(9 * Co)
    wr = torch.randn(Co, nlat + hid, 1, 1, generator=g) / math.sqrt(nlat + hid)
    b0, b1, br = (torch.randn(Co, generator=g) * 0.1 for _ in range(3))
    xa = to_nhwc(x)
    p = {}
    for k, (w, b) in dict(c0=(w0, b0), c1=(w1, b1), res=(wr, br)).items():
        wt, cp = pack_conv(w.to(DEV), rup(w.shape[1], 4))
        p[k] = ConvW(wt, b.to(DEV), Co, w.shape[2], w.shape[3], cp)
    p["cout"] = Co
    lat_dev = dv(msg)
    old = eng.resblock_msg0(xa, p, "tm", lat_dev, 1, nlat)                      # wave-specialised kernels, fp32 out
    want = eng.to_planes(old, "tm.want").clone()
    ghost, xpl = eng.resblock_msg0(xa, p, "tm", lat_dev, 1, nlat, planes_out=True)
    torch.cuda.synchronize()
    assert xpl is not None and ghost.t is None and ghost.C == Co
    n = 2 * old.rows * Co
    got = xpl[:n].clone()
    if B == 32:
        assert eng._planes_split(ghost) == 1
        assert torch.equal(got, want[:n])
    else:
        assert eng._planes_split(ghost) > 1
        def value(pl):      # hi + lo of the [2][C/16][rows][16] f16 planes, back in fp32 units
            h = pl.view(torch.float16).float().view(2, -1)
            return (h[0] + h[1]) / 16.0
        ref = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, b0, padding=1)), w1, b1, padding=1)) + F.conv2d(x, wr, br)
        a, b_ = value(got), value(want[:n])
        assert (a - b_).abs().max().item() < 2e-5 * float(ref.abs().max())
    env = os.environ.get("VIDEOSEAL_MSG0_PLANES")
    eng.msg0_planes = False
    try:
        o2, pl2 = eng.resblock_msg0(xa, p, "tm", lat_dev, 1, nlat, planes_out=True)
        assert pl2 is None and torch.equal(o2.t, old.t)
    finally:
        eng.msg0_planes = env != "0"


def test_msg_latent_and_broadcast(eng):
    g = torch.Generator().manual_seed(10)
    B, k, hid = 3, 40, 24
    table = torch.randn(2 * k, hid, generator=g)
    msgs = torch.randint(0, 2, (B, k), generator=g)
    ref = F.embedding(2 * torch.arange(k)[None] + msgs, table).sum(-2)
    lat = torch.empty(B, hid, device=DEV)
    N.check(eng.lib.vs_msg_latent(N.ptr(dv(table)), N.ptr(dv(msgs.to(torch.int32))), B, k, hid, N.ptr(lat), N.stream()), "ml")
    dst = torch.zeros(B, 6, 32, device=DEV)
    N.check(eng.lib.vs_broadcast_channels(N.ptr(lat), B, hid, N.ptr(dst), B, 6, 32, 8, N.stream()), "bc")
    torch.cuda.synchronize()
    assert (lat.cpu() - ref).abs().max() < 1e-5
    d = dst.cpu()
    assert (d[:, :, 8:] == lat.cpu()[:, None, :]).all() and (d[:, :, :8] == 0).all()


@pytest.mark.parametrize("Cout", [1, 3])
def test_outc_tanh(eng, Cout):
    g = torch.Generator().manual_seed(11)
    B, Cc, H, W = 2, 16, 10, 12
    x = torch.randn(B, Cc, H, W, generator=g)
    w, b = torch.randn(Cout, Cc, 1, 1, generator=g) / 4, torch.randn(Cout, generator=g)
    ref = torch.tanh(F.conv2d(x, w, b))
    xa = to_nhwc(x)
    out = torch.empty(B, Cout, H, W, device=DEV)
    N.check(eng.lib.vs_outc_tanh(N.ptr(xa.t), H * W, B, Cc, xa.ld, N.ptr(dv(w.reshape(Cout, Cc))), N.ptr(dv(b)), Cout, 1,
                                 N.ptr(out), N.stream()), "outc")
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 1e-6


def test_pool_linear(eng):
    g = torch.Generator().manual_seed(12)
    B, Cc, H, W, Nn = 3, 76, 8, 8, 33
    x = torch.randn(B, Cc, H, W, generator=g)
    w, b = torch.randn(Nn, Cc, generator=g) / 8, torch.randn(Nn, generator=g)
    ref = F.linear(x.mean(dim=[-2, -1]), w, b)
    xa = to_nhwc(x)
    out = torch.empty(B, Nn, device=DEV)
    N.check(eng.lib.vs_pool_linear(N.ptr(xa.t), B, H * W, Cc, xa.ld, N.ptr(dv(w)), N.ptr(dv(b)), Nn, N.ptr(out), N.stream()), "pl")
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 1e-5


RESIZE_CASES = [(768, 768, 256, 256, 1), (200, 328, 256, 256, 1), (144, 176, 64, 64, 1), (64, 64, 64, 64, 1),
                (768, 768, 256, 256, 0), (90, 130, 64, 64, 0), (1080, 1920, 256, 256, 1), (100, 60, 256, 256, 1), (31, 45, 64, 64, 0)]


@pytest.mark.parametrize("case", RESIZE_CASES)
def test_resize_pre(eng, case):
    H, W, oh, ow, aa = case
    g = torch.Generator().manual_seed(H + W)
    B = 3
    x = torch.rand(B, 3, H, W, generator=g)
    ref = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False, antialias=bool(aa))
    M0 = torch.tensor([0.299, 0.587, 0.114])
    yref = (torch.matmul(ref.permute(0, 2, 3, 1), M0) * 2 - 1)[::2]
    xd = x.to(DEV)
    rgb = torch.full((B, oh, ow, 4), float("nan"), device=DEV)
    key = torch.full((2, oh, ow, 4), float("nan"), device=DEV)
    ymat = (C.c_float * 3)(0.299, 0.587, 0.114)
    N.check(eng.lib.vs_resize_pre(N.ptr(xd), B, 3, H, W, oh, ow, aa, N.ptr(rgb), 2.0, -1.0, N.ptr(key), 2, ymat, N.stream()), "rs")
    torch.cuda.synchronize()
    got = rgb.cpu()
    assert (got[..., :3].permute(0, 3, 1, 2) - (ref * 2 - 1)).abs().max() < 2e-6     # ATen separable filter restated
    assert (got[..., 3] == 0).all()
    assert (key.cpu()[..., 0] - yref).abs().max() < 2e-6


def test_u8_unit_conversion_is_exact(eng):
    """uint8 -> [0,1] inside the kernels == `torch.tensor(u8, dtype=float32) / 255.0` on the CPU (inference_streaming.py:26) for all 256 values:
    a 1:1 'resize' of a 16 x 16 RGB24 frame holding every byte value returns the conversion itself."""
    u = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1).repeat(1, 1, 1, 3).contiguous()
    ref = u.float() / 255.0
    rgb = torch.empty(1, 16, 16, 4, device=DEV)
    for aa in (0, 1):
        N.check(eng.lib.vs_resize_pre_u8(N.ptr(u.to(DEV)), 1, 16, 16, 16, 16, aa, N.ptr(rgb), 1.0, 0.0, None, 1, None, N.stream()), "rs")
        torch.cuda.synchronize()
        assert torch.equal(rgb.cpu()[..., :3], ref)


def _jnd_ref(x):
    lum = 0.299 * (255 * x[:, 0:1]) + 0.587 * (255 * x[:, 1:2]) + 0.114 * (255 * x[:, 2:3])
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]])[None, None]
    ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]])[None, None]
    kl = torch.tensor([[1., 1., 1., 1., 1.], [1., 2., 2., 2., 1.], [1., 2., 0., 2., 1.], [1., 2., 2., 2., 1.], [1., 1., 1., 1., 1.]])[None, None]
    la = F.conv2d(lum, kl, padding=2) / 32
    la = torch.where(la <= 127, 17 * (1 - torch.sqrt(la / 127 + 1e-5)), 3 / 128 * (la - 127) + 3)
    gx, gy = F.conv2d(lum, kx, padding=1), F.conv2d(lum, ky, padding=1)
    cm = torch.sqrt(gx ** 2 + gy ** 2)
    cm = 0.117 * (16 * cm ** 2.4 / (cm ** 2 + 26 ** 2))
    return torch.clamp_min(la + cm - 0.3 * torch.minimum(la, cm), 0) / 255, torch.cat([kl.flatten(), kx.flatten(), ky.flatten()])


def test_jnd_heatmap_nchw(eng):
    from oracle.inputs import synthetic_frames
    x = synthetic_frames(2, 70, 101, seed=3)
    ref, taps = _jnd_ref(x)
    t43 = (C.c_float * 43)(*[float(v) for v in taps])
    h = torch.empty(2, 1, 70, 101, device=DEV)
    N.check(eng.lib.vs_jnd_heatmap(N.ptr(dv(x)), 2, 70, 101, 3 * 70 * 101, 70 * 101, 101, 1, t43, N.ptr(h), N.stream()), "jnd")
    torch.cuda.synchronize()
    assert (h.cpu() - ref).abs().max() < 2e-6      # heat-maps are O(0.05); powf/sqrtf ulp differences only


@pytest.mark.parametrize("n,cin,kh,kw,arith", [(20, 16, 3, 3, 3), (384, 384, 3, 3, 2), (96, 48, 1, 1, 3), (257, 768, 1, 1, 2), (33, 4, 4, 1, 3)])
def test_device_weight_pack_is_bit_identical_to_the_torch_pack(n, cin, kh, kw, arith):
    """csrc/pack.hip (one launch) against engine.split_f16x2 / split_bf16x3 + pack_blocked"""
    from videoseal_amd import engine as E
    g = torch.Generator().manual_seed(n + cin)
    w = (torch.randn(n, cin, kh, kw, generator=g) * 0.05).cuda()
    wt, cp = E.pack_conv(w, cin)
    cw = E.ConvW(wt, None, n, kh, kw, cp).with_blk(arith)
    # reference on the CPU: torch's CPU float -> half conversion rounds to nearest EVEN like v_cvt_f16_f32 does; the ROCm build's device-side
    # conversion breaks exact ties the other way (a quarter of the low terms are exact ties), so it is not the yardstick
    if arith == 2:
        ref_split, w_mul = E.split_f16x2(wt.cpu())
    else:
        ref_split, w_mul = E.split_bf16x3(wt.cpu()), 1.0
    assert cw.w_mul == w_mul
    assert torch.equal(cw.split.cpu(), ref_split)
    assert torch.equal(cw.blk.cpu(), E.pack_blocked(ref_split, kh * kw))


@pytest.mark.parametrize("co,ci,k,in_ld", [(16, 3, 3, 4), (384, 384, 3, 384), (40, 130, 1, 132), (24, 64, 2, 64), (7, 5, 3, 8)])
def test_pack_conv_on_the_device(co, ci, k, in_ld):
    """vs_pack_conv (one launch) == the ATen formulation of engine.pack_conv, and its transpose mode == pack_conv of the flipped / transposed
    weights (the backward-data weights of training.py)"""
    from videoseal_amd.engine import pack_conv_bwd
    g = torch.Generator().manual_seed(co + ci)
    w = torch.randn(co, ci, k, k, generator=g)
    ref, cp = pack_conv(w, in_ld)                       # CPU tensor: the ATen path
    got, cp2 = pack_conv(w.to(DEV), in_ld)
    assert cp == cp2 and torch.equal(got.cpu(), ref)
    out_ld = (co + 3) // 4 * 4
    wb = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()
    refb, cpb = pack_conv(wb, out_ld)
    gotb, cpb2 = pack_conv_bwd(w.to(DEV), out_ld)
    assert cpb == cpb2 and torch.equal(gotb.cpu(), refb)


@pytest.mark.parametrize("case", ["full_jnd", "full_jnd_fwd_order_preds", "lowres_hmap", "no_attenuation_rgb_delta", "interpolate_mode", "white_noise_at_the_jump"])
def test_embed_tail_forms_are_bit_identical(case):
    """vs_embed_tail's kernels: separable stencils on 16-row tiles (variant 2, round 3's default) and the row-streaming strip kernel (4 = default:
    frame fetched once, luminance ring, running stencil sums) return the same bits -- same expressions, same summation order; the 43-tap form
    (1) agrees to rounding
    (jnd.py:63-108, wam.py:183-197, videoseal.py:303-344).  Ragged sizes (W not a multiple of 256, H not a multiple of 4), strips of several
    heights, every key-frame mode, Cd = 1 / 3, the training forward's order of operations, frames that sit on the la = 127 jump."""
    import os
    from videoseal_amd.native import TailDesc
    L = N.lib()
    g = torch.Generator(device="cuda").manual_seed(11)
    F_, H, W, S = 6, 333, 541, 96
    lo = torch.rand(F_, 3, H // 16 + 1, W // 16 + 1, device="cuda", generator=g)
    x = (0.8 * torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear") + 0.2 * torch.rand(F_, 3, H, W, device="cuda", generator=g)).clamp_(0, 1).contiguous()
    if case == "white_noise_at_the_jump":
        x = (0.498 + 0.01 * torch.randn(F_, 3, H, W, device="cuda", generator=g)).clamp_(0, 1).contiguous()      # 255 x ~ 127: the branch of jnd.py:66-68
    Cd = 3 if case == "no_attenuation_rgb_delta" else 1
    step = 2 if case in ("interpolate_mode", "lowres_hmap") else 1
    nkey = (F_ + step - 1) // step
    delta = (0.3 * torch.randn(nkey, Cd, S, S, device="cuda", generator=g)).contiguous()
    hm = torch.rand(F_, S, S, device="cuda", generator=g).contiguous()
    taps = (C.c_float * 43)(*([1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 2, 0, 2, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1] + [-1, 0, 1, -2, 0, 2, -1, 0, 1] + [1, 2, 1, 0, 0, 0, -1, -2, -1]))
    att = {"full_jnd": 1, "full_jnd_fwd_order_preds": 2, "lowres_hmap": 1, "no_attenuation_rgb_delta": 0, "interpolate_mode": 1, "white_noise_at_the_jump": 1}[case]
    low = case == "lowres_hmap"
    want_pw = case == "full_jnd_fwd_order_preds"
    vm = 2 if case == "interpolate_mode" else (1 if case == "lowres_hmap" else 0)

    def run(variant, strip=None):
        out = torch.full_like(x, -7.0)
        pw = torch.full((F_, Cd, H, W), -7.0, device="cuda") if want_pw else None
        d = TailDesc()
        d.imgs, d.out, d.preds_w = N.ptr(x), N.ptr(out), N.ptr(pw)
        d.delta, d.hmap_lowres, d.taps43 = N.ptr(delta), (N.ptr(hm) if low else None), C.cast(taps, C.c_void_p)
        d.F, d.H, d.W, d.S_h, d.S_w, d.Cd = F_, H, W, S, S, Cd
        d.step, d.video_mode, d.total_key = step, vm, nkey
        d.attenuate, d.clamp, d.antialias = att, 1, 1
        d.scaling_i, d.scaling_w, d.io_u8, d.variant = 1.0, 0.2, 0, variant
        L.vs_debug_set(2, int(strip or 0))              # development switch: strip height of the streaming tail for this call
        try:
            N.check(L.vs_embed_tail(C.byref(d), N.stream()), "vs_embed_tail")
            torch.cuda.synchronize()
        finally:
            L.vs_debug_set(2, 0)
        return out, pw
    ref, ref_pw = run(2)
    assert float(ref.min()) >= 0.0                      # every pixel written
    v1, _ = run(1)                                      # the 43-tap order is a different association of the same stencils: rounding-level difference
    assert (v1 - ref).abs().max().item() <= (1e-6 if case != "white_noise_at_the_jump" else 1e-3)
    for variant, strip in ((4, None), (4, 4), (4, 20), (4, 52), (4, 96), (4, 512)):
        got, got_pw = run(variant, strip)
        assert torch.equal(got, ref), (case, variant, strip, float((got - ref).abs().max()))
        if want_pw:
            assert torch.equal(got_pw, ref_pw), (case, variant, strip)


@pytest.mark.parametrize("arith", [2, 3])
@pytest.mark.parametrize("shape", [(3, 1, 40, 56), (2, 16, 37, 29), (2, 12, 8, 16), (1, 16, 64, 64)])
def test_resblock_thin_fused_kernel(arith, shape):
    """vs_resblock_thin (unet.py:24-39 with eval BatchNorm folded; `inc` and the last `ups` block of VideoSeal 1.0): the whole ResnetBlock in one
    launch, t on chip, v_mfma_f32_16x16x32 with two taps per instruction -- against torch fp32 on the CPU and against the two-launch path of the
    same engine (same arithmetic, different K order: fp32-rounding agreement, 2e-5 of the output range like every conv test here).  Ragged maps
    (not multiples of the 8 x 16 tile), 1 / 12 / 16 input channels, both operand splits."""
    B, Cin, H, W = shape
    e = Eng(arith=arith)
    g = torch.Generator().manual_seed(31 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w0 = torch.randn(16, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b0 = torch.randn(16, generator=g) * 0.3
    w1 = torch.randn(16, 16, 3, 3, generator=g) / math.sqrt(16 * 9)
    b1 = torch.randn(16, generator=g) * 0.3
    wr = torch.randn(16, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    br = torch.randn(16, generator=g) * 0.3
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, b0, padding=1)), w1, b1, padding=1)) + F.conv2d(x, wr, br)
    xa = to_nhwc(x)

    def packed():
        wt0, cp0 = pack_conv(w0.to(DEV), xa.ld)
        wt1, cp1 = pack_conv(w1.to(DEV), 16)
        wtr, cpr = pack_conv(wr.to(DEV), xa.ld)
        return dict(c0=ConvW(wt0, dv(b0), 16, 3, 3, cp0), c1=ConvW(wt1, dv(b1), 16, 3, 3, cp1), res=ConvW(wtr, dv(br), 16, 1, 1, cpr), cout=16)
    p = packed()
    assert e._thin_ok(xa, p, None)
    got = from_nhwc(e.resblock(xa, p, "thin"))
    torch.cuda.synchronize()
    assert rel_err(got, ref) < 2e-5
    e.thin_fused = False
    two = from_nhwc(e.resblock(xa, packed(), "thin2"))
    torch.cuda.synchronize()
    assert rel_err(two, ref) < 2e-5 and rel_err(got, two) < 2e-5


@pytest.mark.parametrize("cfg", [(5, 333, 541, 96, 96, 1, 2), (3, 768, 768, 256, 256, 1, 1), (4, 200, 300, 128, 160, 0, 3), (2, 96, 80, 256, 256, 1, 1),
                                 (3, 256, 256, 256, 256, 0, 1), (2, 700, 1100, 224, 352, 1, 4),
                                 # round 6: the 64-column tile of the streaming kernel (scales up to 4: BASELINE configs[4], 1024 -> 256), a 3.9 : 1 ragged
                                 # shape, and widths that are multiples of four with window starts at every residue (16-byte row pieces)
                                 (2, 1024, 1024, 256, 256, 1, 1), (2, 1000, 996, 256, 256, 1, 2), (3, 500, 600, 160, 192, 1, 1), (2, 772, 764, 250, 260, 1, 1)])
def test_resize_pre_forms_are_bit_identical(cfg):
    """vs_resize_pre (wam.py:161-172 + the RGB -> Y / x*2-1 pre-processing): the row-streaming separable kernel (default: every input row fetched
    once per 128-column tile, filtered horizontally once) against the 32 x 8 tile kernel -- same expressions, same order, same bits; down- and
    up-scales, antialias on / off, ragged sizes, key frames every `step`, rgb / key / both outputs."""
    import os
    B, H, W, oh, ow, aa, step = cfg
    L = N.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(B, 3, H, W, device="cuda", generator=g).contiguous()
    ymat = (C.c_float * 3)(0.299, 0.587, 0.114)
    nk = (B + step - 1) // step

    def run(form, want_rgb=True, want_key=True, y=True):
        rgb = torch.full((B, oh, ow, 4), -9.0, device="cuda") if want_rgb else None
        key = torch.full((nk, oh, ow, 4), -9.0, device="cuda") if want_key else None
        L.vs_debug_set(0, 1 if form == "tile" else 0)   # development switch: kernel form for this call
        try:
            N.check(L.vs_resize_pre(N.ptr(x), B, 3, H, W, oh, ow, aa, N.ptr(rgb), 2.0, -1.0, N.ptr(key), step, ymat if y else None, N.stream()), "vs_resize_pre")
            torch.cuda.synchronize()
        finally:
            L.vs_debug_set(0, 0)
        return rgb, key
    r0, k0 = run("tile")
    r1, k1 = run(None)
    assert float(r0.min()) > -9.0 and float(k0.min()) > -9.0
    assert torch.equal(r0, r1) and torch.equal(k0, k1)
    ref = F.interpolate(x.cpu(), size=(oh, ow), mode="bilinear", align_corners=False, antialias=bool(aa))
    assert (r1[..., :3].permute(0, 3, 1, 2).cpu() - (ref * 2 - 1)).abs().max() < 2e-6
    for kw in (dict(want_rgb=False), dict(want_key=False), dict(y=False)):
        a, b = run("tile", **kw), run(None, **kw)
        assert all((p is None and q is None) or torch.equal(p, q) for p, q in zip(a, b))
    for strip in (2, 6, 512):
        L.vs_debug_set(1, strip)
        try:
            r2, k2 = run(None)
        finally:
            L.vs_debug_set(1, 0)
        assert torch.equal(r2, r0) and torch.equal(k2, k0), strip


@pytest.mark.parametrize("shape", [(2, 32, 37, 29), (3, 32, 16, 48), (1, 32, 128, 128)])
def test_resblock_thin_fused_kernel_32_channels(shape):
    """the 32-channel form of vs_resblock_thin (VideoSeal 1.0's 128^2 level: `downs.0.conv`, `ups.1.conv`; weights in LDS, one tap x 32 channels per
    matrix instruction, 2 x f16 arithmetic): against torch fp32 on the CPU and the two-launch path; ragged maps; the exact 3 x bf16 split keeps the
    two-launch path (its three planes do not fit the kernel's LDS)"""
    B, Cin, H, W = shape
    e = Eng(arith=2)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, Cin, H, W, generator=g)
    w0 = torch.randn(32, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b0 = torch.randn(32, generator=g) * 0.3
    w1 = torch.randn(32, 32, 3, 3, generator=g) / math.sqrt(32 * 9)
    b1 = torch.randn(32, generator=g) * 0.3
    wr = torch.randn(32, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    br = torch.randn(32, generator=g) * 0.3
    ref = F.relu(F.conv2d(F.relu(F.conv2d(x, w0, b0, padding=1)), w1, b1, padding=1)) + F.conv2d(x, wr, br)
    xa = to_nhwc(x)

    def packed():
        wt0, cp0 = pack_conv(w0.to(DEV), xa.ld)
        wt1, cp1 = pack_conv(w1.to(DEV), 32)
        wtr, cpr = pack_conv(wr.to(DEV), xa.ld)
        return dict(c0=ConvW(wt0, dv(b0), 32, 3, 3, cp0), c1=ConvW(wt1, dv(b1), 32, 3, 3, cp1), res=ConvW(wtr, dv(br), 32, 1, 1, cpr), cout=32)
    p = packed()
    assert e._thin_ok(xa, p, None)
    got = from_nhwc(e.resblock(xa, p, "thin32"))
    torch.cuda.synchronize()
    assert rel_err(got, ref) < 2e-5
    e.thin_fused = False
    two = from_nhwc(e.resblock(xa, packed(), "thin32b"))
    torch.cuda.synchronize()
    assert rel_err(two, ref) < 2e-5 and rel_err(got, two) < 2e-5
    e3 = Eng(arith=3)
    assert not e3._thin_ok(xa, packed(), None)


# ---- red zones: kernels that load unconditionally from clamped addresses and select afterwards must not let what lies
# around a tensor reach the result, and must not write outside the output.  Every operand sits between two guard areas poisoned with signalling
# patterns: NaN around the inputs (a value read past either end that reaches an accumulator makes the output NaN or different), a sentinel
# around the output (a store outside the tensor changes it).  The unguarded launch of the same case is the expected value, bit for bit.
GUARD = 4096        # floats on either side (16 KiB: more than any tile's halo)


def _guarded(t: torch.Tensor, fill: float):
    flat = t.contiguous().flatten()
    buf = torch.full((GUARD + flat.numel() + GUARD,), fill, device=t.device, dtype=t.dtype)
    buf[GUARD:GUARD + flat.numel()] = flat
    return buf, buf[GUARD:GUARD + flat.numel()].view(t.shape)


def _guards_intact(buf: torch.Tensor, fill: float) -> bool:
    lo, hi = buf[:GUARD], buf[-GUARD:]
    if fill != fill:
        return bool(torch.isnan(lo).all() and torch.isnan(hi).all())
    return bool((lo == fill).all() and (hi == fill).all())


REDZONE_CONV = [c for c in CONV_CASES if c[-1] in (10, 11, 12, 15, 0x40, 0x43, 0x44, 0x45, 14)]


@pytest.mark.parametrize("case", REDZONE_CONV)
def test_conv_kernels_ignore_and_preserve_what_surrounds_their_tensors(eng, case):
    """patch / wave-specialised / thin-layer conv kernels (clamped unconditional patch loads, conv3x3_patch_pc's round-5 out-of-bounds read was
    of exactly this class): same launch with every operand between poisoned guard areas"""
    B, Cin, H, W, Cout, k, s, p, pm, act, tile = case
    if not eng.use_split:
        pytest.skip("patch / wave-specialised kernels exist for the split back-end only")
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    xa = to_nhwc(x)
    wt, cp = pack_conv(w.to(DEV), xa.ld)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    plain = eng.new_act("rz.o", B, Ho, Wo, Cout)
    plain.t.fill_(3.0)
    eng.conv(xa, ConvW(wt, b.to(DEV), Cout, k, k, cp), plain, stride=s, pad=p, pad_mode=pm, act=act, tile_hint=tile)
    torch.cuda.synchronize()
    want = plain.t.clone()
    xbuf, xg = _guarded(xa.t, float("nan"))
    bbuf, bg = _guarded(b.to(DEV), float("nan"))
    obuf, og = _guarded(torch.full_like(plain.t, 3.0), -7.0)
    _KEEP.extend([xbuf, bbuf, obuf])
    out = Act(og, B, Ho, Wo, Cout, plain.ld)
    eng.conv(Act(xg, B, H, W, Cin, xa.ld), ConvW(wt, bg, Cout, k, k, cp), out, stride=s, pad=p, pad_mode=pm, act=act, tile_hint=tile)
    torch.cuda.synchronize()
    assert torch.isfinite(og).all(), "a value from outside the input tensor reached the output"
    assert torch.equal(og.flatten(), want.flatten())
    assert _guards_intact(obuf, -7.0), "a store outside the output tensor"
    assert _guards_intact(xbuf, float("nan")) and _guards_intact(bbuf, float("nan"))


@pytest.mark.parametrize("C_,H,W", [(96, 16, 16), (24, 9, 13), (362, 7, 7), (192, 40, 36), (20, 64, 64), (130, 17, 50)])
def test_dwconv7_ln_ignores_and_preserves_what_surrounds_its_tensors(eng, C_, H, W):
    """depthwise 7x7 + LayerNorm (halo tiles, ragged channel chunks): the same red-zone harness"""
    g = torch.Generator().manual_seed(C_ + H)
    B = 2
    xa = to_nhwc(torch.randn(B, C_, H, W, generator=g))
    wdw = torch.randn(49, rup(C_, 4), generator=g) * 0.1
    wdw[:, C_:] = 0
    wdw = dv(wdw)
    vecs = [dv(F.pad(torch.randn(C_, generator=g), (0, rup(C_, 4) - C_))) for _ in range(3)]
    L, st = eng.lib, N.stream()

    def run(xt, ot):
        N.check(L.vs_dwconv7_ln(N.ptr(xt), B, H, W, C_, xa.ld, N.ptr(wdw), N.ptr(vecs[0]), N.ptr(vecs[1]), N.ptr(vecs[2]), 1e-6, N.ptr(ot), xa.ld, st),
                "vs_dwconv7_ln")
        torch.cuda.synchronize()
    want = torch.full_like(xa.t, 3.0)
    run(xa.t, want)
    xbuf, xg = _guarded(xa.t, float("nan"))
    obuf, og = _guarded(torch.full_like(xa.t, 3.0), -7.0)
    run(xg, og)
    assert torch.isfinite(og).all() and torch.equal(og, want)
    assert _guards_intact(obuf, -7.0) and _guards_intact(xbuf, float("nan"))


@pytest.mark.parametrize("case", [
    # B, H, W, K, N, tile, split_k                 (partials: K channels x H*W/32 <= 16 rows per frame)
    (3, 16, 16, 1536, 384, 10, 1),                 # ConvNeXt stage-2 pwconv2 on tile 26: 8 partial rows, one frame per 128-row tile
    (5, 8, 8, 3072, 768, 2, 4),                    # stage-3 pwconv2: two frames per tile, 4 K slices (every slice needs the mean over ALL channels)
    (2, 8, 16, 768, 200, 1, 2),                    # 4 partial rows, ragged N
    (2, 16, 32, 256, 96, 10, 1),                   # 16 partial rows (the cap): groups of four
    (3, 8, 24, 384, 130, 1, 1),                    # 6 partial rows: kper = 2, an empty fourth group
])
def test_grn_finish_folded_into_the_gemm_equals_the_separate_launch(eng, case):
    """ABI v3 (round 6): `grn_part / grn_gamma / grn_nchunk` make the wave-specialised 1x1 GEMM derive GRN's scale = 1 + gamma * Gx / (mean Gx + 1e-6)
    (common.py:158-169) from pwconv1's ||h||^2 partials in its own prologue.  Against the path it replaces -- vs_grn_scale_from_partials, then the
    same GEMM reading the scale rows -- the outputs are IDENTICAL bit for bit (same sums in the same order), and equal torch's GRN + linear."""
    if not eng.use_split:
        pytest.skip("split back-end only")
    B, H, W, K, Nn, tl, sk = case
    if tl == 10 and eng.arith != 2:
        pytest.skip("tile 26 exists in the 2 x f16 arithmetic only")
    HW = H * W
    g = torch.Generator().manual_seed(sum(case))
    h = torch.randn(B, HW, K, generator=g) * (0.5 + torch.rand(B, 1, K, generator=g))
    gamma, beta = 0.5 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias, res = torch.randn(Nn, generator=g), torch.randn(B, HW, Nn, generator=g)
    gx = h.double().pow(2).sum(1, keepdim=True).sqrt()
    nx = gx / (gx.mean(-1, keepdim=True) + 1e-6)
    ref = F.linear((gamma.double() * (h.double() * nx) + beta.double() + h.double()).float(), w, bias) + res
    # partials as pwconv1's epilogue writes them: [B][HW / 32][K] sums of squares of 32-row groups
    part = dv(h.pow(2).view(B, HW // 32, 32, K).sum(2))
    ha = Act(dv(h), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    cw = ConvW(wt, dv(bias), Nn, 1, 1, cp)
    ld = (Nn + 3) // 4 * 4
    rr = Act(torch.zeros(B * HW * ld, device=DEV), B, H, W, Nn, ld)
    rr.t.view(B, HW, ld)[..., :Nn] = res.to(DEV)
    gm, bt = dv(gamma), dv(beta)
    outs = []
    for fold in (True, False):
        eng.grn_fold = fold
        scale = torch.full((B * K + 16,), float("nan"), device=DEV)          # never written in fold mode, never read either
        ra = Act(torch.full((B * HW * ld,), float("nan"), device=DEV), B, H, W, Nn, ld)
        try:
            eng.conv(ha, cw, ra, res=rr, tile_hint=N.CONV_TILE_HI | tl, split_k=sk, a_scale=scale, a_scale_ld=K, a_shift=bt,
                     grn_fold=(part, gm, B, HW, K))
        finally:
            eng.grn_fold = True
        torch.cuda.synchronize()
        assert torch.isnan(scale[:B * K]).all() == fold
        full = ra.t.view(B, HW, ld).cpu()
        assert rel_err(full[..., :Nn], ref) < 3e-5
        outs.append(full)
    assert torch.equal(outs[0], outs[1])
    # every other tile code refuses the folded form instead of reading a scale nobody wrote
    d_bad = Act(torch.zeros(B * HW * ld, device=DEV), B, H, W, Nn, ld)
    with pytest.raises(N.NativeError):
        eng.grn_fold = False
        try:
            import ctypes as C_
            orig = eng.lib.vs_conv_gemm

            def poke(dref, st):
                dref._obj.grn_part, dref._obj.grn_gamma, dref._obj.grn_nchunk = N.ptr(part), N.ptr(gm), HW // 32
                return orig(dref, st)
            eng.lib.vs_conv_gemm = poke
            eng.conv(ha, cw, d_bad, res=rr, tile_hint=1, a_scale=dv(torch.ones(B * K + 16)), a_scale_ld=K, a_shift=bt)
        finally:
            eng.lib.vs_conv_gemm = orig
            eng.grn_fold = True


@pytest.mark.parametrize("rows,C,ld,hw,grn", [(961 * 3, 5792, 5792, 961, True), (64, 128, 128, 32, True), (33, 16, 20, 33, False), (450, 1472, 1472, 225, True),
                                               (1000, 144, 160, 250, True), (31, 48, 48, 31, True)])
def test_to_planes_affine_forms_are_bit_identical(rows, C, ld, hw, grn):
    """vs_to_planes_affine (GRN apply + f16 operand split of h in front of the planes pwconv2, gemm_pl.hip): the LDS-transposed kernel (round 6:
    512-byte row pieces in, 1 KiB runs out) against the one-row-per-wave kernel (development switch 5) and against the split computed on the host:
    ChunkySeal's stage-2 shape (31 x 31 frames, 5 792 channels = 45.25 channel tiles), ragged row tiles, channel strides above C, fewer than 32 rows"""
    L = N.lib()
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, ld, generator=g) * 3
    B = (rows + hw - 1) // hw
    scale = (1 + 0.3 * torch.randn(B, C, generator=g)).contiguous()
    shift = (0.1 * torch.randn(C, generator=g)).contiguous()
    xd, sd, hd = dv(x), dv(scale), dv(shift)
    outs = []
    for form in (0, 1):
        pl = torch.full((2 * rows * C,), 0x7777, dtype=torch.int16, device=DEV)
        L.vs_debug_set(5, form)
        try:
            N.check(L.vs_to_planes_affine(N.ptr(xd), rows, C, ld, 1.0, N.ptr(sd) if grn else None, C, N.ptr(hd) if grn else None, hw, N.ptr(pl), N.stream()),
                    "to_planes_affine")
            torch.cuda.synchronize()
        finally:
            L.vs_debug_set(5, 0)
        outs.append(pl.cpu())
    assert torch.equal(outs[0], outs[1])
    # planes [2][C / 16][rows][16]: hi = f16(v), lo = f16(v - hi) of v = x * scale[frame] + shift (fp32)
    v = x[:, :C]
    if grn:
        v = torch.addcmul(shift[None], v, scale[torch.arange(rows) // hw])        # (one rounding of the product + sum or two: tolerance below)
    hi = outs[0][:rows * C].view(torch.float16).view(C // 16, rows, 16).permute(1, 0, 2).reshape(rows, C).float()
    lo = outs[0][rows * C:].view(torch.float16).view(C // 16, rows, 16).permute(1, 0, 2).reshape(rows, C).float()
    assert ((hi + lo) - v).abs().max().item() <= 2e-6 * max(1.0, v.abs().max().item())


@pytest.mark.parametrize("B,H,W,K,Nn", [(3, 15, 15, 160, 200), (2, 31, 31, 96, 200), (5, 7, 9, 64, 40), (16, 31, 31, 48, 724), (1, 6, 6, 32, 70)])
def test_grn_statistics_from_straddling_row_groups(B, H, W, K, Nn):
    """round 6 (ChunkySeal's 31 x 31 frames): `vs_conv_desc_t::sumsq_hw` makes the planes GEMM's epilogue write, per 32-row group, the sums of
    squares of its rows in the frame of its first row | in the next frame ([rows/32][2][N]); vs_grn_scale_from_straddle_partials adds per frame the
    slots that belong to it.  Against common.py:166-168 evaluated on the stored fp32 activations, and against vs_grn_scale (the separate pass over
    h it replaces); the GEMM's output is untouched by the option; other tile codes refuse it."""
    eng = Eng(arith=2)
    HW = H * W
    g = torch.Generator().manual_seed(B * HW + Nn)
    x = torch.randn(B, HW, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias, gamma = torch.randn(Nn, generator=g), 0.5 * torch.randn(Nn, generator=g)
    xa = Act(dv(x), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    cw = ConvW(wt, dv(bias), Nn, 1, 1, cp)
    pl = torch.empty(B * HW * K * 2, dtype=torch.int16, device=DEV)
    N.check(eng.lib.vs_to_planes(N.ptr(xa.t), xa.rows, K, xa.ld, 16.0, N.ptr(pl), N.stream()), "to_planes")
    ld = (Nn + 3) // 4 * 4
    outs = []
    for straddle in (True, False):
        out = Act(torch.full((B * HW * ld,), float("nan"), device=DEV), B, H, W, Nn, ld)
        parts = torch.full((((B * HW + 31) // 32) * 2 * Nn,), float("nan"), device=DEV)
        kw = dict(sumsq=parts, sumsq_hw=HW) if straddle else {}
        eng.conv(xa, cw, out, act=N.ACT_GELU, tile_hint=N.CONV_TILE_HI | 8, in_pl=pl, a_mul=16.0, arith=2, **kw)
        outs.append((out, parts))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0].t, outs[1][0].t)
    h = outs[0][0].t.view(B, HW, ld)[..., :Nn]
    gm = dv(gamma)
    if HW >= 32:
        scale = torch.full((B, ld), float("nan"), device=DEV)
        N.check(eng.lib.vs_grn_scale_from_straddle_partials(N.ptr(outs[0][1]), B, HW, Nn, N.ptr(gm), N.ptr(scale), ld, N.stream()), "straddle finish")
        scale2 = torch.full((B, ld), float("nan"), device=DEV)
        part = torch.empty(((HW + 63) // 64) * B * Nn, device=DEV)
        N.check(eng.lib.vs_grn_scale(N.ptr(outs[0][0].t), B, HW, Nn, ld, N.ptr(gm), N.ptr(part), N.ptr(scale2), N.stream()), "vs_grn_scale")
        torch.cuda.synchronize()
        gx = h.double().pow(2).sum(1).sqrt()
        ref = 1 + gamma.double().to(DEV) * gx / (gx.mean(-1, keepdim=True) + 1e-6)
        assert (scale[:, :Nn].double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
        assert (scale[:, :Nn] - scale2[:, :Nn]).abs().max().item() < 1e-5 * ref.abs().max().item()
        assert (scale[:, Nn:] == 0).all()
    else:      # frames shorter than a row group could straddle three frames: refused
        with pytest.raises(N.NativeError):
            N.check(eng.lib.vs_grn_scale_from_straddle_partials(N.ptr(outs[0][1]), B, HW, Nn, N.ptr(gm), N.ptr(outs[0][1]), ld, N.stream()), "HW < 32")
    with pytest.raises(N.NativeError):      # the generic kernel has no such epilogue
        eng.conv(xa, cw, outs[1][0], act=N.ACT_GELU, tile_hint=1, sumsq=outs[1][1], sumsq_hw=max(HW, 32), arith=2)


@pytest.mark.parametrize("C_,H,W", [(1448, 31, 31), (1030, 9, 5), (96, 6, 7), (2896, 15, 15), (362, 7, 7)])
def test_dwconv7_ln_two_row_strips_equal_single_rows(C_, H, W):
    """round 6: the one-row depthwise 7x7 + LayerNorm kernel with strips of 4 x 2 output pixels (ChunkySeal's 1448- / 2896-channel maps: 10 instead of
    17.5 input fetches per output) against single rows (development switch 6): fp32 rows and operand planes agree to fp32 rounding, odd heights (a half strip at the bottom)"""
    L = N.lib()
    g = torch.Generator().manual_seed(C_ + H)
    B = 2
    ld = rup(C_, 32) if C_ >= 128 else rup(C_, 4)
    xa = to_nhwc(torch.randn(B, C_, H, W, generator=g), ld)
    wdw = torch.zeros(49, ld); wdw[:, :C_] = torch.randn(49, C_, generator=g) * 0.1
    wdw = dv(wdw)
    vecs = [dv(F.pad(torch.randn(C_, generator=g), (0, ld - C_))) for _ in range(3)]
    Cp = rup(C_, 32)
    outs = []
    import os
    for rows in (2, 1):
        L.vs_debug_set(6, rows)
        try:
            o = torch.full((B * H * W * ld,), 3.0, device=DEV)
            pl = torch.full((2 * B * H * W * Cp,), 77, dtype=torch.int16, device=DEV)
            # VS_DWCONV=0 is process-wide; these shapes take the one-row kernel anyway (tile + LayerNorm buffer beyond the LDS) except the small ones
            N.check(L.vs_dwconv7_ln(N.ptr(xa.t), B, H, W, C_, ld, N.ptr(wdw), N.ptr(vecs[0]), N.ptr(vecs[1]), N.ptr(vecs[2]), 1e-6, N.ptr(o), ld, N.stream()), "dw")
            N.check(L.vs_dwconv7_ln_planes(N.ptr(xa.t), B, H, W, C_, ld, N.ptr(wdw), N.ptr(vecs[0]), N.ptr(vecs[1]), N.ptr(vecs[2]), 1e-6, 16.0, Cp, N.ptr(pl),
                                           N.stream()), "dw planes")
            torch.cuda.synchronize()
        finally:
            L.vs_debug_set(6, 0)
        outs.append((o.cpu(), pl.cpu()))
    # same depthwise sums (same tap order per output); the LayerNorm behind them spreads a pixel's channels over more or fewer lanes depending on the
    # pixels per workgroup, so the two forms agree to fp32 rounding
    a, b = outs[0][0].view(B * H * W, ld), outs[1][0].view(B * H * W, ld)
    assert torch.isfinite(a).all() and (a - b).abs().max().item() < 2e-6 * max(1.0, b.abs().max().item())
    def deq(pl):
        hi = pl[:B * H * W * Cp].view(torch.float16).float()
        lo = pl[B * H * W * Cp:].view(torch.float16).float()
        return (hi + lo) / 16.0
    assert (deq(outs[0][1]) - deq(outs[1][1])).abs().max().item() < 2e-6 * max(1.0, b.abs().max().item())
    # and the planes are the split of the fp32 rows
    full = torch.zeros(B * H * W, Cp)
    full[:, :C_] = a[:, :C_]
    got = deq(outs[0][1]).view(Cp // 16, B * H * W, 16).permute(1, 0, 2).reshape(B * H * W, Cp)
    assert (got - full).abs().max().item() < 2e-6 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("case", [
    # B, H, W, K, N, act, res, split_k, sumsq (0 / 1 / HW = straddling groups)
    (2, 16, 16, 96, 384, 2, False, 1, 1),          # pwconv1-like: GELU + GRN partials, 2 x 1.5 tiles
    (3, 15, 15, 160, 520, 0, True, 1, 0),          # ragged rows (675) and columns, residual
    (2, 31, 31, 48, 724, 2, False, 1, 961),        # ChunkySeal-like: straddling GRN partials
    (16, 31, 31, 64, 1448, 0, True, 1, 0),         # 61 x 6 tiles: the XCD-grouped order
    (1, 8, 8, 3072, 768, 0, True, 8, 0),           # K slices + epilogue kernel
    (5, 8, 8, 16, 40, 1, False, 1, 0),             # a single K step, one ragged tile
    (2, 16, 24, 48, 300, 3, False, 1, 0),          # three K steps (odd count), tanh
])
def test_gemm_planes_big_tile(case):
    """tile code 27 (round 6): the all-DMA planes GEMM on 256 x 256 tiles with one wave per SIMD (128 x 128 per wave, rotating fragment banks, four
    32 KiB LDS stages) multiplies the same products in the same K order as tile 24 -> identical outputs and GRN partials, bit for bit"""
    B, H, W, K, Nn, act, res, sk, sumsq = case
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(41 + K + Nn)
    HW = H * W
    x = torch.randn(B, HW, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    r = torch.randn(B, HW, Nn, generator=g)
    ref = F.linear(x, w, bias)
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref), 3: torch.tanh(ref)}[act]
    if res:
        ref = ref + r
    xa = Act(dv(x), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    cw = ConvW(wt, dv(bias), Nn, 1, 1, cp)
    ld = (Nn + 3) // 4 * 4
    ra = Act(torch.zeros(B * HW * ld, device=DEV), B, H, W, Nn, ld)
    ra.t.view(B, HW, ld)[..., :Nn] = r.to(DEV)
    pl = torch.empty(B * HW * K * 2, dtype=torch.int16, device=DEV)
    N.check(eng.lib.vs_to_planes(N.ptr(xa.t), xa.rows, K, xa.ld, 16.0, N.ptr(pl), N.stream()), "to_planes")
    outs = []
    # (tanh: tile 24's 128 x 192 register tile does not carry the tanh variant any more -- the 128-column tile 25 is the twin there)
    for t in (N.CONV_TILE_HI | 11, N.CONV_TILE_HI | (9 if act == 3 else 8)):
        out = Act(torch.full((B * HW * ld,), float("nan"), device=DEV), B, H, W, Nn, ld)
        part = torch.full((((B * HW + 31) // 32) * 2 * Nn,), float("nan"), device=DEV) if sumsq else None
        kw = dict(sumsq=part, sumsq_hw=(sumsq if sumsq > 1 else 0)) if sumsq else {}
        eng.conv(xa, cw, out, act=act, res=(ra if res else None), tile_hint=t, in_pl=pl, split_k=sk, a_mul=16.0, arith=2, **kw)
        torch.cuda.synchronize()
        got = out.t.view(B, HW, ld)
        assert rel_err(got[..., :Nn].cpu(), ref) < 2e-5
        assert (got[..., Nn:] == 0).all()
        outs.append((out.t.clone(), part))
    assert torch.equal(outs[0][0], outs[1][0])
    if sumsq:
        n = ((B * HW + 31) // 32) * (2 if sumsq > 1 else 1) * Nn
        assert torch.equal(outs[0][1][:n], outs[1][1][:n])


@pytest.mark.parametrize("tile", [2, 8])
def test_wide_register_tiles_refuse_tanh(tile):
    """round 6: the TN = 3 register tiles (tile codes 18 and 24, 96 accumulators per lane at 128 x 192) are built without the tanh epilogue -- it was
    what their epilogues spilled registers for, and no layer of a shipped card takes it there (the only tanh is the 1-channel output conv).  An explicit
    request is refused with VS_ERR_UNSUPPORTED and leaves the output untouched; with K slices (the activation then runs in the epilogue kernel) and
    on every narrower tile tanh still works (GEMM_PL_CASES / GEMM_PC_CASES), and the automatic choice never lands on the refused form."""
    B, H, W, K, Nn = 2, 16, 24, 64, 300
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H * W, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    ref = torch.tanh(F.linear(x, w, bias))
    xa = Act(dv(x), B, H, W, K, K)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    cw = ConvW(wt, dv(bias), Nn, 1, 1, cp)
    pl = torch.empty(B * H * W * K * 2, dtype=torch.int16, device=DEV)
    N.check(eng.lib.vs_to_planes(N.ptr(xa.t), xa.rows, K, xa.ld, 16.0, N.ptr(pl), N.stream()), "to_planes")
    kw = dict(in_pl=pl, a_mul=16.0) if tile == 8 else {}
    out = Act(torch.full((B * H * W * Nn,), float("nan"), device=DEV), B, H, W, Nn, Nn)
    with pytest.raises(N.NativeError, match="code -2"):
        eng.conv(xa, cw, out, act=3, tile_hint=N.CONV_TILE_HI | tile, arith=2, **kw)
    torch.cuda.synchronize()
    assert torch.isnan(out.t).all()
    for hint, sk in ((N.CONV_TILE_HI | tile, 2), (0, 1)):         # K slices on the same tile; the dispatcher's own choice
        eng.conv(xa, cw, out, act=3, tile_hint=hint, split_k=sk, arith=2, **kw)
        torch.cuda.synchronize()
        assert rel_err(out.t.view(B, H * W, Nn).cpu(), ref) < 2e-5


def _guard_i16(t: torch.Tensor):
    """int16 operand planes between two areas of f16 NaNs (0x7E00)"""
    flat = t.contiguous().flatten()
    buf = torch.full((2 * GUARD + flat.numel() + 2 * GUARD,), 0x7E00, device=t.device, dtype=torch.int16)
    buf[2 * GUARD:2 * GUARD + flat.numel()] = flat
    return buf, buf[2 * GUARD:2 * GUARD + flat.numel()]


@pytest.mark.parametrize("case", [c for c in GEMM_PC_CASES if c[9] == 1] + [c for c in GEMM_PC_CASES if c[9] > 1][:2])
def test_wave_specialised_gemm_ignores_and_preserves_what_surrounds_its_tensors(eng, case):
    """red zones around the operands of gemm1x1_pc (tile codes 17 / 18 / 26): activations, GRN scale rows, residual and bias between NaN guards, the
    output between sentinels -- the ragged last tiles re-read clamped rows / column groups whose values must never reach a stored element"""
    if not eng.use_split:
        pytest.skip("split back-end only")
    B, H, W, K, Nn, act, grn, use_res, tl, sk = case
    if tl == 10 and eng.arith != 2:
        pytest.skip("tile 26 exists in the 2 x f16 arithmetic only")
    g = torch.Generator().manual_seed(hash(case) % 1000)
    HW = H * W
    h = torch.randn(B, HW, K, generator=g)
    sc = (1 + 0.3 * torch.randn(B, K, generator=g)).contiguous()
    sh = (0.1 * torch.randn(K, generator=g)).contiguous()
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    ld = (Nn + 3) // 4 * 4
    res = torch.zeros(B, HW, ld)
    res[..., :Nn] = torch.randn(B, HW, Nn, generator=g)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)

    def run(guarded):
        keep = []
        def place(t, fill):
            if not guarded:
                return dv(t)
            buf, view = _guarded(t.to(DEV), fill)
            keep.append((buf, fill))
            return view
        hd, scd, shd, bd, rd = place(h, float("nan")), place(sc, float("nan")), place(sh, float("nan")), place(bias, float("nan")), place(res, float("nan"))
        od = place(torch.full((B * HW * ld,), 3.0), -7.0)
        kw = dict(act=act)
        if grn:
            kw.update(a_scale=scd, a_scale_ld=K, a_shift=shd)
        eng.conv(Act(hd, B, H, W, K, K), ConvW(wt, bd, Nn, 1, 1, cp), Act(od, B, H, W, Nn, ld), res=(Act(rd, B, H, W, Nn, ld) if use_res else None),
                 tile_hint=N.CONV_TILE_HI | tl, split_k=sk, **kw)
        torch.cuda.synchronize()
        for buf, fill in keep:
            assert _guards_intact(buf, fill)
        return od.clone()
    want, got = run(False), run(True)
    assert torch.isfinite(got).all() and torch.equal(got, want)


@pytest.mark.parametrize("case", [c for c in GEMM_PL_CASES if c[9] == 1][:7] + [GEMM_PL_CASES[3]])
def test_planes_gemm_ignores_and_preserves_what_surrounds_its_tensors(case):
    """the same for the all-DMA planes GEMM (tile codes 24 / 25 and the one-wave-per-SIMD tile 27): operand planes between f16 NaNs, residual / bias between
    fp32 NaNs, output and GRN partials between sentinels"""
    B, H, W, K, Nn, act, grn, res, tile, sk, sumsq = case
    eng = Eng(arith=2)
    g = torch.Generator().manual_seed(77)
    HW = H * W
    x = torch.randn(B, HW, K, generator=g)
    w = torch.randn(Nn, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Nn, generator=g)
    r = torch.randn(B, HW, Nn, generator=g)
    wt, cp = pack_conv(w[:, :, None, None].to(DEV), K)
    pl0 = torch.empty(B * HW * K * 2, dtype=torch.int16, device=DEV)
    xd = dv(x)
    N.check(eng.lib.vs_to_planes(N.ptr(xd), B * HW, K, K, 16.0, N.ptr(pl0), N.stream()), "to_planes")
    torch.cuda.synchronize()
    for t in (tile, 27):
        outs = []
        for guarded in (False, True):
            keep = []
            def place(tt, fill):
                if not guarded:
                    return dv(tt)
                buf, view = _guarded(tt.to(DEV), fill)
                keep.append((buf, fill))
                return view
            if guarded:
                pbuf, pl = _guard_i16(pl0)
            else:
                pbuf, pl = None, pl0
            bd, rd = place(bias, float("nan")), place(r.contiguous(), float("nan"))
            od = place(torch.full((B * HW * Nn,), 3.0), -7.0)
            part = place(torch.full((((B * HW + 31) // 32) * Nn,), 5.0), -7.0) if sumsq else None
            eng.conv(Act(xd, B, H, W, K, K), ConvW(wt, bd, Nn, 1, 1, cp), Act(od, B, H, W, Nn, Nn), act=act, res=(Act(rd, B, H, W, Nn, Nn) if res else None),
                     tile_hint=N.CONV_TILE_HI | (t - 16), in_pl=pl, split_k=sk, sumsq=part, a_mul=16.0, arith=2)
            torch.cuda.synchronize()
            for buf, fill in keep:
                assert _guards_intact(buf, fill)
            if pbuf is not None:
                assert (pbuf[:2 * GUARD] == 0x7E00).all() and (pbuf[-2 * GUARD:] == 0x7E00).all()
            outs.append((od.clone(), part.clone() if part is not None else None))
        assert torch.isfinite(outs[1][0]).all() and torch.equal(outs[0][0], outs[1][0])
        if sumsq:
            assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("C_,ld,rows,act", [(96, 96, 4099, 0), (192, 192, 1000, 2), (384, 384, 257, 0), (768, 768, 130, 2), (724, 736, 63 * 5, 0), (1448, 1472, 97, 0),
                                            (2896, 2912, 33, 2), (20, 20, 77, 1), (100, 128, 50, 0)])
def test_layernorm_lanes_form_equals_the_wave_per_row_form(C_, ld, rows, act):
    """vs_layernorm_act, round 6: LPP lanes per row with the row in registers (8 lanes for 96 channels ... 64 lanes x 12 float4 for ChunkySeal's 2896)
    against the one-wave-per-row kernel (development switch 7) and torch: the mean / variance sums run in another order -> fp32 rounding; pad lanes of
    the output (ld > C) are written as zeros by both"""
    L = N.lib()
    g = torch.Generator().manual_seed(C_ + rows)
    x = torch.zeros(rows, ld)
    x[:, :C_] = torch.randn(rows, C_, generator=g) * 2 + 0.5
    w, b = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.1
    xd, wd, bd = dv(x), dv(w), dv(b)
    outs = []
    for form in (0, 1):
        o = torch.full((rows, ld), 9.0, device=DEV)
        L.vs_debug_set(7, form)
        try:
            N.check(L.vs_layernorm_act(N.ptr(xd), rows, C_, ld, N.ptr(wd), N.ptr(bd), 1e-6, act, N.ptr(o), ld, N.stream()), "ln")
            torch.cuda.synchronize()
        finally:
            L.vs_debug_set(7, 0)
        outs.append(o.cpu())
    ref = F.layer_norm(x[:, :C_], (C_,), w, b, 1e-6)
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref)}[act]
    for o in outs:
        assert (o[:, :C_] - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
        assert (o[:, C_:] == 0).all()
    assert (outs[0] - outs[1]).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,W,C_", [(3, 8, 12, 96), (2, 9, 7, 192), (2, 16, 16, 384), (1, 2, 2, 8), (2, 5, 6, 724)])
def test_layernorm_into_the_patch_matrix_of_the_downsampling_conv(B, H, W, C_):
    """vs_layernorm_patch2x2 (round 6): LayerNorm of every pixel written as the patch matrix [B][H/2][W/2][4C] of the 2 x 2 / stride-2 conv behind it
    (convnext.py:109-117), tap-major (ky, kx): equals vs_layernorm_act followed by the rearrangement (bit for bit: same kernel, other store address);
    odd maps drop the last row / column like the conv; and the 1x1 GEMM on it with the conv's packed weights equals F.conv2d(stride 2)"""
    L = N.lib()
    g = torch.Generator().manual_seed(H * W + C_)
    x = torch.randn(B, C_, H, W, generator=g) * 2
    w, b = torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.1
    xa = to_nhwc(x, C_)
    wd, bd = dv(w), dv(b)
    flat = torch.empty(B * H * W * C_, device=DEV)
    N.check(L.vs_layernorm_act(N.ptr(xa.t), B * H * W, C_, C_, N.ptr(wd), N.ptr(bd), 1e-6, 0, N.ptr(flat), C_, N.stream()), "ln")
    Ho, Wo = H // 2, W // 2
    pat = torch.full((B * Ho * Wo * 4 * C_,), float("nan"), device=DEV)
    N.check(L.vs_layernorm_patch2x2(N.ptr(xa.t), B, H, W, C_, C_, N.ptr(wd), N.ptr(bd), 1e-6, N.ptr(pat), N.stream()), "ln patch")
    torch.cuda.synchronize()
    ln = flat.view(B, H, W, C_)[:, :2 * Ho, :2 * Wo]
    want = ln.reshape(B, Ho, 2, Wo, 2, C_).permute(0, 1, 3, 2, 4, 5).reshape(B, Ho, Wo, 4 * C_)
    if C_ > 64:
        assert torch.equal(pat.view(B, Ho, Wo, 4 * C_), want)
    else:       # (vs_layernorm_act takes its thread-per-row kernel for C <= 64: other summation order)
        assert (pat.view(B, Ho, Wo, 4 * C_) - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item())
    if C_ % 8 == 0 and Ho * Wo > 0:
        from videoseal_amd.engine import pack_patch_conv
        eng = Eng(arith=2)
        Co = 40
        cw_ = torch.randn(Co, C_, 2, 2, generator=g) / math.sqrt(4 * C_)
        cb = torch.randn(Co, generator=g)
        wt, cp = pack_patch_conv(cw_.to(DEV), C_)
        out = eng.new_act("dp.o", B, Ho, Wo, Co)
        eng.conv(Act(pat, B, Ho, Wo, 4 * C_, 4 * C_), ConvW(wt, dv(cb), Co, 1, 1, 2 * cp), out)
        torch.cuda.synchronize()
        ref = F.conv2d(F.layer_norm(x.permute(0, 2, 3, 1), (C_,), w, b, 1e-6).permute(0, 3, 1, 2), cw_, cb, stride=2)
        assert rel_err(from_nhwc(out), ref) < 3e-5


@pytest.mark.parametrize("stride,Co,S,B", [(4, 96, 64, 3), (2, 96, 37, 2), (4, 64, 20, 1), (4, 128, 256, 2), (2, 128, 12, 5)])
def test_stem_conv_and_layernorm_in_one_kernel(stride, Co, S, B):
    """vs_stem_conv_ln (round 6): the ConvNeXt stem -- 4 x 4 patchify conv + LayerNorm (convnext.py:100-104) -- as exact fp32 multiply-adds on the vector
    ALUs, four lanes per pixel: against torch (conv2d + layer_norm) and against the two launches it replaces (split-operand GEMM + vs_layernorm_act)"""
    L = N.lib()
    g = torch.Generator().manual_seed(stride + Co + S)
    x = torch.rand(B, 3, S, S + 4, generator=g) * 2 - 1
    w = torch.randn(Co, 3, 4, 4, generator=g) / 7
    b = torch.randn(Co, generator=g) * 0.2
    lw, lb = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    ref = F.layer_norm(F.conv2d(x, w, b, stride=stride).permute(0, 2, 3, 1), (Co,), lw, lb, 1e-6)
    xa = to_nhwc(x, 4)
    wt, cp = pack_patch_conv(w.to(DEV), 4)
    Ho, Wo = (S - 4) // stride + 1, (S + 4 - 4) // stride + 1
    out = torch.full((B, Ho, Wo, Co), float("nan"), device=DEV)
    bd, lwd, lbd = dv(b), dv(lw), dv(lb)
    N.check(L.vs_stem_conv_ln(N.ptr(xa.t), B, S, S + 4, stride, N.ptr(wt), N.ptr(bd), N.ptr(lwd), N.ptr(lbd), 1e-6, Co, N.ptr(out), Co, N.stream()), "stem")
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())
    eng = Eng(arith=2)
    t = eng.new_act("stem2.c", B, Ho, Wo, Co)
    eng.conv(xa, ConvW(wt, bd, Co, 4, 1, cp), t, geom=(Wo, stride * 4, 16, stride, 1, 0, 0))
    two = torch.empty(B * Ho * Wo * Co, device=DEV)
    N.check(L.vs_layernorm_act(N.ptr(t.t), B * Ho * Wo, Co, t.ld, N.ptr(lwd), N.ptr(lbd), 1e-6, 0, N.ptr(two), Co, N.stream()), "ln")
    torch.cuda.synchronize()
    assert (out.flatten() - two).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    # unsupported widths are refused, not mangled
    assert L.vs_stem_conv_ln(N.ptr(xa.t), B, S, S + 4, stride, N.ptr(wt), N.ptr(bd), N.ptr(lwd), N.ptr(lbd), 1e-6, 100, N.ptr(out), 100, N.stream()) == N.ERR_UNSUPPORTED
