"""Backward of the detector fine-tuning step (train.py:517-523 + 626-643) on the HIP path.

Unit level: every kernel of csrc/bwd_ops.hip against torch autograd of the same op on the GPU (fp32, tolerances written per test).
End to end: `videoseal_amd.training.DetectorStep` against the gradients the UNMODIFIED reference produced with its own `VideosealLoss`
and `loss.backward()` (tests/golden/make_golden_bwd.py; the oracle reproduces the same fixtures on the CPU in tests/test_oracle_bwd.py)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.weights import make_state_dict, tiny_spec  # noqa: E402
from tests._util import load_golden, projection_vector  # noqa: E402
from tests.test_gpu_e2e import make_model  # noqa: E402

from videoseal_amd import native as N  # noqa: E402


def _lib():
    return N.lib(), N.stream()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _padded(t, ld):
    """[rows, C] -> [rows, ld] with zero pad lanes"""
    out = torch.zeros(t.shape[0], ld, device=t.device)
    out[:, : t.shape[1]] = t
    return out


@pytest.mark.parametrize("rows,n,k,ldn,ldk", [(1000, 20, 36, 20, 36), (4100, 96, 384, 96, 384), (64, 257 - 1, 16, 256, 16), (7, 5, 70, 8, 72),
                                              (2500, 130, 200, 132, 200), (16384, 384, 3456, 384, 3456), (33, 64, 64, 64, 64), (5000, 70, 1000, 72, 1000)])
def test_gemm_wgrad(rows, n, k, ldn, ldk):
    L, st = _lib()
    dy, x = _padded(_rand(rows, n, seed=1), ldn), _padded(_rand(rows, k, seed=2), ldk)
    part = torch.empty(int(L.vs_gemm_wgrad_partial_floats(rows, n, k)), device="cuda")
    dw = torch.empty(n, k, device="cuda")
    N.check(L.vs_gemm_wgrad(N.ptr(dy), ldn, n, N.ptr(x), ldk, k, rows, N.ptr(part), N.ptr(dw), st), "vs_gemm_wgrad")
    ref = dy[:, :n].double().t() @ x[:, :k].double()
    assert (dw.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    dw2 = torch.empty_like(dw)
    N.check(L.vs_gemm_wgrad(N.ptr(dy), ldn, n, N.ptr(x), ldk, k, rows, N.ptr(part), N.ptr(dw2), st), "vs_gemm_wgrad")
    assert torch.equal(dw, dw2)          # deterministic


@pytest.mark.parametrize("B,H,W,ci,co,stride,reflect", [
    (2, 12, 10, 64, 64, 1, 0), (3, 9, 11, 70, 130, 1, 0), (2, 16, 16, 128, 64, 2, 0), (2, 15, 13, 64, 96, 2, 0), (1, 33, 17, 384, 384, 1, 0),
    (16, 32, 32, 384, 384, 1, 0),
    # the thin outer levels (register-tile kernel): inc, down0, ups of the U-Net and odd shapes
    (2, 20, 18, 16, 16, 1, 0), (1, 256, 256, 3, 16, 1, 0), (2, 33, 31, 16, 32, 2, 0), (3, 40, 24, 32, 32, 1, 0), (2, 64, 64, 12, 8, 1, 0),
    (4, 128, 128, 16, 32, 2, 0), (1, 5, 3, 4, 4, 1, 0),
    # reflection padding: the Upsample convs (cat of the up-sampled map and the skip: 48 -> 16 @256^2, 96 -> 32, 512 -> 128)
    (2, 64, 48, 48, 16, 1, 1), (2, 40, 40, 96, 32, 1, 1), (2, 16, 24, 512, 128, 1, 1), (1, 2, 2, 8, 4, 1, 1), (1, 19, 35, 20, 12, 1, 1)])
def test_conv3x3_wgrad_from_the_image(B, H, W, ci, co, stride, reflect):
    """vs_conv3x3_wgrad (matrix-core kernel on the implicit patch matrix / register-tile kernel) against the fp64 weight gradient of
    F.conv2d(stride, padding=1, zeros | reflect), in the [co][tap * ld + c] layout of the patch-matrix route it replaces
    (unet.py:21-27 double_conv, :52 the stride-2 down conv, :170-197 the Upsample conv)"""
    L, st = _lib()
    ld = (ci + 3) // 4 * 4
    x = _rand(B, ci, H, W, seed=21)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = _rand(B, co, Ho, Wo, seed=23)
    xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect") if reflect else F.pad(x.double(), (1, 1, 1, 1))
    cols = F.unfold(xp, 3, padding=0, stride=stride)                            # [B, ci * 9, Ho * Wo]
    wgrad = torch.einsum("bnl,bkl->nk", dy.double().reshape(B, co, -1), cols).reshape(co, ci, 3, 3)     # = autograd of F.conv2d w.r.t. the weight
    xa = _padded(x.permute(0, 2, 3, 1).reshape(-1, ci), ld).contiguous()
    dya = dy.permute(0, 2, 3, 1).reshape(-1, co).contiguous()
    ldn = (co + 3) // 4 * 4
    dya = _padded(dya, ldn)
    assert L.vs_conv3x3_wgrad_supported(co, ld, stride)
    part = torch.empty(int(L.vs_conv3x3_wgrad_partial_floats(co, ld, B, H, W, stride)), device="cuda")
    dw = torch.full((co, 9 * ld), 7.0, device="cuda")
    pm = N.PAD_REFLECT if reflect else N.PAD_ZERO
    N.check(L.vs_conv3x3_wgrad(N.ptr(dya), ldn, co, N.ptr(xa), ld, B, H, W, stride, pm, N.ptr(part), N.ptr(dw), st), "vs_conv3x3_wgrad")
    got = dw.view(co, 3, 3, ld)[..., :ci].permute(0, 3, 1, 2)
    assert (got.double() - wgrad).abs().max() <= 2e-5 * wgrad.abs().max()
    assert (dw.view(co, 9, ld)[..., ci:] == 0).all()
    dw2 = torch.empty_like(dw)
    N.check(L.vs_conv3x3_wgrad(N.ptr(dya), ldn, co, N.ptr(xa), ld, B, H, W, stride, pm, N.ptr(part), N.ptr(dw2), st), "vs_conv3x3_wgrad")
    assert torch.equal(dw, dw2)          # deterministic


@pytest.mark.parametrize("B,H,W,C,ld", [(2, 6, 7, 8, 8), (1, 2, 2, 4, 4), (3, 3, 9, 10, 12), (1, 64, 48, 48, 48)])
def test_pad_embed_and_reflect_fold_are_the_adjoint_of_reflection_padding(B, H, W, C, ld):
    """backward data of a reflection-padded conv = zero-padded conv over the padded map + vs_reflect_fold1; checked against autograd of F.pad"""
    L, st = _lib()
    x = _rand(B, C, H, W, seed=31).requires_grad_(True)
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    dxp = _rand(B, C, H + 2, W + 2, seed=32)
    xp.backward(dxp)
    dxpa = _padded(dxp.permute(0, 2, 3, 1).reshape(-1, C), ld).contiguous()
    out = torch.full((B * H * W, ld), 7.0, device="cuda")
    N.check(L.vs_reflect_fold1(N.ptr(dxpa), B, H, W, ld, N.ptr(out), st), "vs_reflect_fold1")
    ref = x.grad.permute(0, 2, 3, 1).reshape(-1, C)
    assert (out[:, :C] - ref).abs().max() <= 1e-6 * ref.abs().max()
    dy = _padded(_rand(B * H * W, C, seed=33), ld)
    canvas = torch.full((B * (H + 2) * (W + 2), ld), 7.0, device="cuda")
    N.check(L.vs_pad_embed1(N.ptr(dy), B, H, W, ld, N.ptr(canvas), st), "vs_pad_embed1")
    cv = canvas.view(B, H + 2, W + 2, ld)
    assert torch.equal(cv[:, 1:-1, 1:-1], dy.view(B, H, W, ld))
    assert (cv[:, 0] == 0).all() and (cv[:, -1] == 0).all() and (cv[:, :, 0] == 0).all() and (cv[:, :, -1] == 0).all()


@pytest.mark.parametrize("B,H,W,C,ld", [(2, 9, 11, 6, 8), (3, 16, 16, 24, 24), (1, 5, 4, 4, 4), (1, 3, 2, 4, 4), (36, 64, 9, 8, 8), (2, 8, 23, 96, 96)])
def test_dwconv7_forward_flip_and_wgrad(B, H, W, C, ld):
    L, st = _lib()
    x = _rand(B, C, H, W, seed=3).requires_grad_(True)
    w = _rand(C, 1, 7, 7, seed=4, scale=0.2).requires_grad_(True)
    b = _rand(C, seed=5).requires_grad_(True)
    y = F.conv2d(x, w, b, padding=3, groups=C)
    dy = _rand(B, C, H, W, seed=6)
    y.backward(dy)

    def nhwc(t):
        return _padded(t.detach().permute(0, 2, 3, 1).reshape(-1, C), ld).contiguous()
    wk = torch.zeros(49, ld, device="cuda")
    wk[:, :C] = w.detach().reshape(C, 49).t()
    bk = torch.zeros(ld, device="cuda")
    bk[:C] = b.detach()
    xa, dya = nhwc(x), nhwc(dy)
    out = torch.zeros(B * H * W, ld, device="cuda")
    N.check(L.vs_dwconv7(N.ptr(xa), B, H, W, C, ld, N.ptr(wk), N.ptr(bk), 0, None, 0, N.ptr(out), ld, st), "vs_dwconv7")
    assert (out[:, :C] - nhwc(y)[:, :C]).abs().max() < 2e-5
    assert (out[:, C:] == 0).all()
    # backward-data = flipped taps; `add` carries the residual branch
    res = _padded(_rand(B * H * W, C, seed=7), ld)
    dx = torch.zeros_like(out)
    N.check(L.vs_dwconv7(N.ptr(dya), B, H, W, C, ld, N.ptr(wk), None, 1, N.ptr(res), ld, N.ptr(dx), ld, st), "vs_dwconv7")
    assert (dx[:, :C] - res[:, :C] - nhwc(x.grad)[:, :C]).abs().max() < 2e-5
    # weight gradient
    part = torch.empty(int(L.vs_dwconv7_wgrad_partial_floats(B, H, ld)), device="cuda")
    dw = torch.empty(49, ld, device="cuda")
    N.check(L.vs_dwconv7_wgrad(N.ptr(xa), ld, N.ptr(dya), ld, B, H, W, C, N.ptr(part), N.ptr(dw), st), "vs_dwconv7_wgrad")
    ref = w.grad.reshape(C, 49).t()
    assert (dw[:, :C] - ref).abs().max() <= 2e-5 * ref.abs().max()


@pytest.mark.parametrize("rows,C,ld", [(37, 16, 16), (1000, 96, 96), (300, 18, 20), (5, 130, 160), (700, 1100, 1100)])
def test_layernorm_bwd(rows, C, ld):
    L, st = _lib()
    x = (_rand(rows, C, seed=8) * 2 + 0.3).requires_grad_(True)
    w, b = _rand(C, seed=9).requires_grad_(True), _rand(C, seed=10).requires_grad_(True)
    y = F.layer_norm(x, (C,), w, b, 1e-6)
    dy = _rand(rows, C, seed=11)
    y.backward(dy)
    xa, dya = _padded(x.detach(), ld), _padded(dy, ld)
    dx = torch.full((rows, ld), 7.0, device="cuda")
    stats = torch.empty(2 * rows, device="cuda")
    part = torch.empty(int(L.vs_colreduce_partial_floats(1, rows, ld)), device="cuda")
    dw, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    N.check(L.vs_layernorm_bwd(N.ptr(xa), ld, N.ptr(dya), ld, N.ptr(w.detach()), rows, C, 1e-6, N.ptr(dx), ld, N.ptr(stats), N.ptr(part),
                               N.ptr(dw), N.ptr(db), st), "vs_layernorm_bwd")
    assert (dx[:, :C] - x.grad).abs().max() <= 2e-5 * x.grad.abs().max()
    assert (dx[:, C:] == 0).all()
    assert (dw - w.grad).abs().max() <= 2e-5 * w.grad.abs().max()
    assert (db - b.grad).abs().max() <= 2e-5 * b.grad.abs().max()


@pytest.mark.parametrize("B,HW,C,ld", [(3, 20, 8, 8), (2, 300, 64, 64), (4, 64, 24, 32), (1, 1000, 12, 12), (2, 333, 1536, 1536)])
def test_gelu_grn_bwd(B, HW, C, ld):
    L, st = _lib()
    h1 = _rand(B, HW, C, seed=12).requires_grad_(True)
    gamma, beta = _rand(C, seed=13).requires_grad_(True), _rand(C, seed=14).requires_grad_(True)
    h2 = F.gelu(h1)
    gx = torch.norm(h2, p=2, dim=1, keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    h3 = gamma * (h2 * nx) + beta + h2
    d3 = _rand(B, HW, C, seed=15)
    h3.backward(d3)
    h1a, d3a = _padded(h1.detach().reshape(-1, C), ld), _padded(d3.reshape(-1, C), ld)
    part = torch.empty(int(L.vs_colreduce_partial_floats(B, HW, ld)), device="cuda")
    coef = torch.empty(6 * B * ld, device="cuda")
    dh1 = torch.full((B * HW, ld), 7.0, device="cuda")
    dg, dbt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    N.check(L.vs_gelu_grn_bwd(N.ptr(h1a), ld, N.ptr(d3a), ld, N.ptr(gamma.detach()), B, HW, C, N.ptr(part), N.ptr(coef), N.ptr(dh1), ld,
                              N.ptr(dg), N.ptr(dbt), st), "vs_gelu_grn_bwd")
    ref = h1.grad.reshape(-1, C)
    assert (dh1[:, :C] - ref).abs().max() <= 3e-5 * ref.abs().max()
    assert (dh1[:, C:] == 0).all()
    assert (dg - gamma.grad).abs().max() <= 3e-5 * gamma.grad.abs().max()
    assert (dbt - beta.grad).abs().max() <= 3e-5 * beta.grad.abs().max()


@pytest.mark.parametrize("B,H,W,C,ld,P", [(2, 8, 12, 3, 4, 4), (2, 6, 10, 20, 20, 2), (1, 7, 9, 6, 8, 2)])
def test_patchify_unpatch_are_the_patch_conv_and_its_adjoint(B, H, W, C, ld, P):
    L, st = _lib()
    Ho, Wo = H // P, W // P
    CP = (P * ld + 15) // 16 * 16
    x = _rand(B, C, H, W, seed=16).requires_grad_(True)
    w = _rand(5, C, P, P, seed=17).requires_grad_(True)
    y = F.conv2d(x, w, stride=P)
    dy = _rand(*y.shape, seed=18)
    y.backward(dy)
    xa = _padded(x.detach().permute(0, 2, 3, 1).reshape(-1, C), ld).contiguous()
    cols = torch.full((B * Ho * Wo, P * CP), 7.0, device="cuda")
    N.check(L.vs_patchify(N.ptr(xa), B, H, W, ld, P, N.ptr(cols), st), "vs_patchify")
    wk = torch.zeros(5, P, CP, device="cuda")                                            # engine.pack_patch_conv's k order
    tmp = torch.zeros(5, P, P, ld, device="cuda")
    tmp[..., :C] = w.detach().permute(0, 2, 3, 1)
    wk[:, :, : P * ld] = tmp.reshape(5, P, P * ld)
    wk = wk.reshape(5, P * CP)
    yk = cols @ wk.t()
    assert (yk - y.detach().permute(0, 2, 3, 1).reshape(-1, 5)).abs().max() < 2e-5
    dcols = dy.permute(0, 2, 3, 1).reshape(-1, 5) @ wk                                   # [rows_out][P * CP]
    dx = torch.full((B * H * W, ld), 7.0, device="cuda")
    N.check(L.vs_unpatch(N.ptr(dcols.contiguous()), B, H, W, ld, P, N.ptr(dx), st), "vs_unpatch")
    ref = x.grad.permute(0, 2, 3, 1).reshape(-1, C)
    assert (dx[:, :C] - ref).abs().max() < 2e-5
    assert (dx[:, C:] == 0).all()


@pytest.mark.parametrize("B,H,W,C,ld", [(2, 5, 7, 6, 8), (1, 2, 2, 4, 4), (2, 8, 8, 16, 16), (1, 3, 9, 4, 4)])
def test_col2im3x3_reflect_is_the_adjoint_of_im2col(B, H, W, C, ld):
    L, st = _lib()
    x = _rand(B, C, H, W, seed=19).requires_grad_(True)
    w = _rand(7, C, 3, 3, seed=20).requires_grad_(True)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
    dy = _rand(*y.shape, seed=21)
    y.backward(dy)
    xa = _padded(x.detach().permute(0, 2, 3, 1).reshape(-1, C), ld).contiguous()
    cols = torch.empty(B * H * W, 9 * ld, device="cuda")
    N.check(L.vs_im2col3x3(N.ptr(xa), B, H, W, ld, N.PAD_REFLECT, N.ptr(cols), st), "vs_im2col3x3")
    wk = torch.zeros(7, 9, ld, device="cuda")
    wk[:, :, :C] = w.detach().permute(0, 2, 3, 1).reshape(7, 9, C)
    wk = wk.reshape(7, 9 * ld)
    assert (cols @ wk.t() - y.detach().permute(0, 2, 3, 1).reshape(-1, 7)).abs().max() < 3e-5
    dcols = (dy.permute(0, 2, 3, 1).reshape(-1, 7) @ wk).contiguous()
    dx = torch.full((B * H * W, ld), 7.0, device="cuda")
    N.check(L.vs_col2im3x3_reflect(N.ptr(dcols), B, H, W, ld, N.ptr(dx), st), "vs_col2im3x3_reflect")
    ref = x.grad.permute(0, 2, 3, 1).reshape(-1, C)
    assert (dx[:, :C] - ref).abs().max() < 3e-5


def test_head_pieces_and_the_decoding_loss():
    L, st = _lib()
    B, HW, C, ld, k = 3, 12, 10, 12, 9
    z = _rand(B, HW, C, seed=22).requires_grad_(True)
    lw, lb = _rand(k + 1, C, seed=23).requires_grad_(True), _rand(k + 1, seed=24).requires_grad_(True)
    msgs = (torch.rand(B, k, generator=torch.Generator().manual_seed(25)) > 0.5).cuda()
    T, gs = 1.7, 0.5
    logits = F.linear(F.gelu(z).mean(1), lw, lb)
    loss = F.binary_cross_entropy_with_logits(logits[:, 1:] / T, msgs.float(), reduction="none").mean()
    (loss * gs).backward()
    # loss + d logits
    dl = torch.empty(B, k + 1, device="cuda")
    lv = torch.empty(1, device="cuda")
    N.check(L.vs_bce_logits(N.ptr(logits.detach().contiguous()), N.ptr(msgs.to(torch.int32).contiguous()), B, B, k, T, gs, N.ptr(dl), N.ptr(lv), st),
            "vs_bce_logits")
    assert abs(float(lv) - float(loss.detach())) < 1e-6
    pooled_ref = F.gelu(z.detach()).mean(1)
    # d logits against the closed form gs / T * (sigmoid(z) - m) / (B k), column 0 = 0
    want = torch.zeros(B, k + 1, device="cuda")
    want[:, 1:] = gs / T * (torch.sigmoid(logits.detach()[:, 1:] / T) - msgs.float()) / (B * k)
    assert (dl - want).abs().max() < 1e-7
    # pooled mean, the three small products, GELU backward through the mean
    hl = _padded(F.gelu(z.detach()).reshape(-1, C), ld)
    pooled = torch.empty(B, ld, device="cuda")
    N.check(L.vs_colmean(N.ptr(hl), B, HW, ld, N.ptr(pooled), st), "vs_colmean")
    assert (pooled[:, :C] - pooled_ref).abs().max() < 1e-6
    dlt = dl.t().contiguous()
    dlw = torch.empty(k + 1, C, device="cuda")
    N.check(L.vs_matmul_small(N.ptr(dlt), B, N.ptr(pooled), ld, k + 1, C, B, N.ptr(dlw), C, st), "vs_matmul_small")
    assert (dlw - lw.grad).abs().max() <= 2e-5 * lw.grad.abs().max()
    dlb = torch.empty(k + 1, device="cuda")
    N.check(L.vs_matmul_small(N.ptr(torch.ones(B, device="cuda")), B, N.ptr(dl), k + 1, 1, k + 1, B, N.ptr(dlb), k + 1, st), "vs_matmul_small")
    assert (dlb - lb.grad).abs().max() <= 2e-5 * lb.grad.abs().max()
    dpooled = torch.zeros(B, ld, device="cuda")
    N.check(L.vs_matmul_small(N.ptr(dl), k + 1, N.ptr(lw.detach().contiguous()), C, B, C, k + 1, N.ptr(dpooled), ld, st), "vs_matmul_small")
    za = _padded(z.detach().reshape(-1, C), ld)
    dz = torch.full((B * HW, ld), 7.0, device="cuda")
    N.check(L.vs_pool_gelu_bwd(N.ptr(za), ld, N.ptr(dpooled), ld, B, HW, C, N.ptr(dz), ld, st), "vs_pool_gelu_bwd")
    ref = z.grad.reshape(-1, C)
    assert (dz[:, :C] - ref).abs().max() <= 3e-5 * ref.abs().max()
    assert (dz[:, C:] == 0).all()


# ---------------------------------------------------------------------------------------------------------- end to end
def _train_forward(model, spec, meta):
    from tests.test_gpu_fwd import hip_forward
    full = dict(meta, bn_train=True, video_mode="repeat", lowres=False, scaling_i=spec.scaling_i)
    return hip_forward(model, spec, full)


@pytest.mark.parametrize("name", ["tiny_bwd_img_recipe", "tiny_bwd_img_balanced", "tiny_bwd_vid_recipe"])
def test_detector_step_matches_the_reference_backward(name):
    from videoseal_amd.training import DetectorStep
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    g = load_golden(name)
    meta = g["meta"]
    model = make_model(spec, sd)
    out, imgs, msgs = _train_forward(model, spec, meta)
    assert out["selected_aug"] == meta["selected_aug"]
    T = meta["temperature"]
    gold_preds = torch.from_numpy(g["preds"]).cuda()                          # already divided by T (train.py:628)
    assert (out["preds"] / T - gold_preds).abs().max() < 1e-3
    # the detector's share of the step: the decoding term with the scale the reference logged (adaptive or fixed), / accumulation
    gscale = meta["log"]["scale_decode"] / meta["accumulation"]
    for p in model.parameters():
        p.grad = None
    loss, logits, grads = DetectorStep(model).step(out["imgs_aug"], out["msgs"], temperature=T, grad_scale=gscale)
    torch.cuda.synchronize()
    assert (logits - out["preds"]).abs().max() < 2e-4                         # the operand-keeping forward == the fused inference forward
    assert abs(float(loss) - meta["log"]["loss_decode"]) < 2e-4
    names = [str(k) for k in g["grad_names"]]
    ref = g["grad_summary"]
    gmax = max(ref[i, 0] for i, k in enumerate(names) if k.startswith("detector."))
    params = dict(model.named_parameters())
    checked = 0
    for i, k in enumerate(names):
        if not k.startswith("detector."):
            assert params[k].grad is None, k                                  # the embedder is frozen
            continue
        gr = params[k].grad
        assert gr is not None, k
        gd = gr.double().flatten().cpu()
        got = np.array([float(gd.norm()), float(gd.sum()), float((gd * projection_vector(k, gd.numel())).sum())])
        tol = 3e-3 * max(ref[i, 0], 1e-3 * gmax) * max(1.0, math.sqrt(gd.numel()) / 16)
        assert np.all(np.abs(got - ref[i]) <= tol), (k, got, ref[i], tol)
        checked += 1
    assert checked == sum(k.startswith("detector.") for k in names) > 50
    for k in [str(k) for k in g["full_names"]]:
        if k.startswith("detector."):
            rf = torch.from_numpy(g["grad." + k]).cuda()
            assert (params[k].grad - rf).abs().max() <= 3e-3 * rf.abs().max() + 1e-9, k
    # a second call accumulates (train.py zeroes the gradients once per batch of accumulation steps)
    first = {k: params[k].grad.clone() for k in names if k.startswith("detector.")}
    DetectorStep(model).step(out["imgs_aug"], out["msgs"], temperature=T, grad_scale=gscale)
    for k, v in first.items():
        assert (params[k].grad - 2 * v).abs().max() <= 1e-6 * v.abs().max() + 1e-12, k


def test_detector_step_full_size_architecture_matches_oracle_autograd():
    """VideoSeal 1.0's extractor (dims 96..768, depths 3-3-9-3, 4C = 3072 channel rows: the multi-sweep paths of the column reductions,
    the 64 x 64 tiling of the weight-gradient GEMM on real shapes) on 2 frames: every detector gradient against torch autograd through the
    oracle's functional forward (CPU, fp32), element-wise."""
    import os
    from oracle import loss as OL
    from oracle import videoseal_ref as R
    from oracle.inputs import synthetic_frames, synthetic_msgs
    from oracle.weights import spec_from_card
    from tests.test_oracle_golden import CARDS
    from videoseal_amd.training import DetectorStep
    spec = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
    sd = make_state_dict(spec, seed=0)
    imgs = synthetic_frames(2, spec.img_size, spec.img_size, seed=77)
    msgs = synthetic_msgs(2, spec.nbits, seed=77)
    names = [k for k, v in sd.items() if k.startswith("detector.") and v.dtype.is_floating_point]
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sdg[k].requires_grad_(True)
    preds = R.extractor_forward(sdg, spec, imgs)
    loss_ref = OL.decoding_loss(preds, msgs, None)
    loss_ref.backward()
    model = make_model(spec, sd)
    loss, logits, grads = DetectorStep(model).step(imgs.cuda(), msgs, accumulate=False)
    torch.cuda.synchronize()
    assert (logits.cpu() - preds.detach()).abs().max() < 2e-4
    assert abs(float(loss) - float(loss_ref.detach())) < 1e-5
    assert set(grads) == set(names)
    worst = 0.0
    for k in names:
        ref = sdg[k].grad
        got = grads[k].reshape(ref.shape).cpu()
        err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        worst = max(worst, err)
        assert err < 3e-3, (k, err)
    print(f"worst relative gradient error over {len(names)} tensors: {worst:.2e}")


# ---------------------------------------------------------------------------------------------------------- ViT extractor (legacy card)
@pytest.mark.parametrize("rows,C,ld", [(100, 20, 20), (37, 130, 132), (1000, 1536, 1536)])
def test_gelu_bwd(rows, C, ld):
    L, st = _lib()
    z = _rand(rows, C, seed=41).requires_grad_(True)
    dy = _rand(rows, C, seed=42)
    F.gelu(z).backward(dy)
    dz = torch.full((rows, ld), 7.0, device="cuda")
    za, dya = _padded(z.detach(), ld), _padded(dy, ld)          # (kept alive: the C-ABI only sees raw pointers)
    N.check(L.vs_gelu_bwd(N.ptr(za), ld, N.ptr(dya), ld, rows, C, N.ptr(dz), ld, st), "vs_gelu_bwd")
    assert (dz[:, :C] - z.grad).abs().max() <= 2e-6 * z.grad.abs().max() + 1e-7
    assert (dz[:, C:] == 0).all()


@pytest.mark.parametrize("cfg", [(2, 16, 16, 6, 64, 0), (2, 16, 16, 6, 64, 8), (3, 8, 8, 2, 16, 4), (1, 8, 12, 3, 32, 0), (2, 8, 8, 2, 16, 0),
                                 (1, 16, 8, 2, 32, 8), (2, 4, 4, 1, 16, 0)])
def test_vit_attention_bwd(cfg):
    """vs_vit_attention_bwd against torch autograd of vit.py:302-360 (scaled q.k^T + decomposed relative positions, softmax, @v) incl. the
    window partition: d qkv and the two relative-position tables"""
    L, st = _lib()
    B, H, W, heads, hd, win = cfg
    D = heads * hd
    g = torch.Generator().manual_seed(43)
    qkv = torch.randn(B, H, W, 3 * D, generator=g).cuda().requires_grad_(True)
    Th, Tw = (win, win) if win else (H, W)
    rel_h = (0.3 * torch.randn(2 * Th - 1, hd, generator=g)).cuda().requires_grad_(True)
    rel_w = (0.3 * torch.randn(2 * Tw - 1, hd, generator=g)).cuda().requires_grad_(True)
    x = qkv
    if win:
        x = x.view(B, H // win, win, W // win, win, 3 * D).permute(0, 1, 3, 2, 4, 5).reshape(-1, win, win, 3 * D)
    Bw = x.shape[0]
    q, k, v = x.reshape(Bw, Th * Tw, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, Bw * heads, Th * Tw, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    ih = (torch.arange(Th)[:, None] - torch.arange(Th)[None, :] + Th - 1).cuda()
    iw = (torch.arange(Tw)[:, None] - torch.arange(Tw)[None, :] + Tw - 1).cuda()
    rq = q.reshape(Bw * heads, Th, Tw, hd)
    attn = (attn.view(Bw * heads, Th, Tw, Th, Tw) + torch.einsum("bhwc,hkc->bhwk", rq, rel_h[ih])[:, :, :, :, None]
            + torch.einsum("bhwc,wkc->bhwk", rq, rel_w[iw])[:, :, :, None, :]).view(Bw * heads, Th * Tw, Th * Tw)
    o = (attn.softmax(-1) @ v).view(Bw, heads, Th, Tw, hd).permute(0, 2, 3, 1, 4).reshape(Bw, Th, Tw, D)
    if win:
        o = o.view(B, H // win, W // win, win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, D)
    do = torch.randn(B, H, W, D, generator=g).cuda()
    o.backward(do)
    out = torch.empty(B, H, W, D, device="cuda")
    qd, rh, rw = qkv.detach().contiguous(), rel_h.detach().contiguous(), rel_w.detach().contiguous()
    N.check(L.vs_vit_attention(N.ptr(qd), B, H, W, heads, hd, win, N.ptr(rh), N.ptr(rw), N.ptr(out), st), "attn")
    dqkv = torch.full((B, H, W, 3 * D), 7.0, device="cuda")
    scr = torch.empty(int(L.vs_vit_attention_bwd_scratch_floats(B, H, W, heads, win)), device="cuda")
    drh, drw = torch.full_like(rh, 7.0), torch.full_like(rw, 7.0)
    N.check(L.vs_vit_attention_bwd(N.ptr(qd), N.ptr(out), N.ptr(do), B, H, W, heads, hd, win, N.ptr(rh), N.ptr(rw), N.ptr(dqkv), N.ptr(scr), N.ptr(drh),
                                   N.ptr(drw), st), "vs_vit_attention_bwd")
    torch.cuda.synchronize()
    assert (dqkv - qkv.grad).abs().max() <= 2e-4 * qkv.grad.abs().max()
    assert (drh - rel_h.grad).abs().max() <= 2e-4 * rel_h.grad.abs().max()
    assert (drw - rel_w.grad).abs().max() <= 2e-4 * rel_w.grad.abs().max()
    dq2 = torch.empty_like(dqkv)
    N.check(L.vs_vit_attention_bwd(N.ptr(qd), N.ptr(out), N.ptr(do), B, H, W, heads, hd, win, N.ptr(rh), N.ptr(rw), N.ptr(dq2), N.ptr(scr), N.ptr(drh),
                                   N.ptr(drw), st), "vs_vit_attention_bwd")
    assert torch.equal(dqkv, dq2)          # deterministic


@pytest.mark.parametrize("which", ["tiny", "card"])
def test_detector_step_vit_extractor_matches_oracle_autograd(which):
    """The detector fine-tuning step for `extractor: sam` (the legacy videoseal_0.0 card; vit.py:14-144): every `detector.*` gradient -- patch
    embedding, absolute and relative position tables, attention / MLP / LayerNorm weights of every block, neck, pixel decoder -- against torch
    autograd through the oracle's functional forward (CPU, fp32).  'tiny': 8 x 8 tokens, 4 x 4 windows + global blocks, 16-wide heads;
    'card': the released architecture (16 x 16 tokens, windows of 8, 64-wide heads) on two frames.  The position tables are zero-initialised
    in the reference (vit.py:66-69, 334-336): seeded random values here so that their terms take part."""
    import os
    from oracle import loss as OL
    from oracle import videoseal_ref as R
    from oracle.inputs import synthetic_frames, synthetic_msgs
    from oracle.weights import legacy_tiny_spec, spec_from_card
    from tests.test_oracle_golden import CARDS
    from videoseal_amd.training import DetectorStep
    spec = legacy_tiny_spec() if which == "tiny" else spec_from_card(os.path.join(CARDS, "videoseal_0.0.yaml"))
    sd = make_state_dict(spec, seed=5)
    gen = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith(("pos_embed", "rel_pos_h", "rel_pos_w")):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=gen)
    nfr = 3 if which == "tiny" else 2
    imgs = synthetic_frames(nfr, spec.img_size, spec.img_size, seed=78)
    msgs = synthetic_msgs(nfr, spec.nbits, seed=78)
    names = [k for k, v in sd.items() if k.startswith("detector.") and v.dtype.is_floating_point]
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sdg[k].requires_grad_(True)
    preds = R.extractor_forward(sdg, spec, imgs)
    loss_ref = OL.decoding_loss(preds, msgs, None)
    loss_ref.backward()
    model = make_model(spec, sd)
    loss, logits, grads = DetectorStep(model).step(imgs.cuda(), msgs, accumulate=False)
    torch.cuda.synchronize()
    assert (logits.cpu() - preds.detach()).abs().max() < 2e-4
    assert abs(float(loss) - float(loss_ref.detach())) < 1e-5
    assert set(grads) == set(names), (set(names) - set(grads), set(grads) - set(names))
    worst = 0.0
    for k in names:
        ref = sdg[k].grad
        got = grads[k].reshape(ref.shape).cpu()
        err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        worst = max(worst, err)
        assert err < 3e-3, (k, err)
    print(f"worst relative gradient error over {len(names)} tensors: {worst:.2e}")
