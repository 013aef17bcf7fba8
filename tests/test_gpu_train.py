"""The differentiable forward and the generator-side training step on the HIP path (videoseal_amd/autograd.py,
training.GeneratorStep, csrc/bwd_shell.hip).

Unit level: every adjoint kernel of bwd_shell.hip against torch autograd of the reference operator (oracle functions on the CPU, fp32).
End to end: the UNMODIFIED inner loop of train.py:626-643 -- `outputs = model(imgs, masks, msgs)`, `outputs["preds"] /= temperature`,
the loss with `last_layer = model.embedder.get_last_layer()`, `loss.backward()` -- against the gradients the unmodified reference produced
with its own `VideosealLoss` (tests/golden/make_golden_bwd.py: 161 trainable tensors of the tiny architecture in three cases, 339 of
VideoSeal 1.0), and two optimizer steps against autograd through the oracle.

Tolerances: the extractor is smooth (GELU / LayerNorm / GRN) -- its 71 / 201 tensors are held to 3e-3 on norm, sum and a seeded projection.
The U-Net is piecewise linear in ~10^7 ReLU decisions; a forward that differs in the last bits flips a few and moves the gradient
discretely (the CPU oracle in fp32 vs fp64: worst 1.3 %, tests/tools/relu_flip_sensitivity.py).  Its tensors are therefore compared twice: against
the reference fixture with a bound of that order, and against the oracle's autograd run with the HIP forward's own ReLU decisions, tightly."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import augment as OA  # noqa: E402
from oracle import loss as OL  # noqa: E402
from oracle import videoseal_ref as R  # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs  # noqa: E402
from oracle.weights import make_state_dict, spec_from_card, tiny_spec  # noqa: E402
from tests._util import load_golden, projection_vector  # noqa: E402
from tests.test_gpu_bwd import _rand  # noqa: E402
from tests.test_gpu_bwd_unet import _MaskedF, hip_relu_masks  # noqa: E402
from tests.test_gpu_e2e import make_model  # noqa: E402
from tests.test_oracle_golden import CARDS  # noqa: E402

from videoseal_amd import augmentation as G  # noqa: E402
from videoseal_amd import autograd as AG  # noqa: E402
from videoseal_amd import native as N  # noqa: E402
from videoseal_amd.training import GeneratorStep  # noqa: E402

AA = {"mode": "bilinear", "align_corners": False, "antialias": True}


# ------------------------------------------------------------------------------------------------ adjoint kernels
@pytest.mark.parametrize("H,W,oh,ow,aa", [(72, 88, 64, 64, True), (64, 64, 72, 88, True), (64, 64, 72, 88, False), (50, 37, 50, 37, True),
                                          (96, 80, 31, 17, True), (20, 24, 200, 130, False), (8, 8, 3, 5, False), (40, 40, 100, 60, True)])
def test_resize_adjoint(H, W, oh, ow, aa):
    x = _rand(2, 3, H, W, seed=1).cpu().requires_grad_(True)
    y = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False, antialias=aa)
    dy = _rand(2, 3, oh, ow, seed=2)
    y.backward(dy.cpu())
    xg = x.detach().cuda().requires_grad_(True)
    yg = G.resize(xg, (oh, ow), aa)
    assert yg.requires_grad and (yg.detach().cpu() - y.detach()).abs().max() < 2e-6
    yg.backward(dy)
    assert (xg.grad.cpu() - x.grad).abs().max() <= 3e-6 * max(1.0, float(x.grad.abs().max()))


@pytest.mark.parametrize("H,W,i,j,h,w,flip", [(20, 30, 3, 5, 10, 12, False), (20, 30, 0, 0, 20, 30, True), (9, 7, 2, 1, 4, 5, True)])
def test_crop_flip_adjoint(H, W, i, j, h, w, flip):
    x = _rand(2, 3, H, W, seed=3).requires_grad_(True)
    y = G.crop_flip(x, i, j, h, w, flip)
    ref = x.detach()[..., i:i + h, j:j + w]
    assert torch.equal(y.detach(), ref.flip(-1) if flip else ref)
    dy = _rand(2, 3, h, w, seed=4)
    y.backward(dy)
    want = torch.zeros_like(x)
    want[..., i:i + h, j:j + w] = dy.flip(-1) if flip else dy
    assert torch.equal(x.grad, want)


@pytest.mark.parametrize("op,factor", [("brightness", 1.4), ("brightness", 0.5), ("contrast", 1.5), ("contrast", 0.5), ("saturation", 1.5),
                                       ("saturation", 0.3), ("grayscale", 0.0)])
def test_color_adjoints(op, factor):
    x0 = synthetic_frames(3, 40, 56, seed=5)
    xc = x0.clone().requires_grad_(True)
    fn = {"brightness": OA.brightness, "contrast": OA.contrast, "saturation": OA.saturation}
    y = OA.grayscale(xc) if op == "grayscale" else fn[op](xc, factor)
    dy = _rand(3, 3, 40, 56, seed=6)
    y.backward(dy.cpu())
    xg = x0.cuda().requires_grad_(True)
    yg = G.color_op(xg, op, factor)
    assert (yg.detach().cpu() - y.detach()).abs().max() < 1e-5
    yg.backward(dy)
    # pixels whose blend lands within rounding of the clamp bounds may pass on one side only: compare away from them
    safe = ((y.detach() > 1e-5) & (y.detach() < 1 - 1e-5)).all(1, keepdim=True).expand_as(y) if op != "grayscale" else torch.ones_like(y, dtype=torch.bool)
    err = (xg.grad.cpu() - xc.grad).abs()
    assert err[safe].max() < 2e-5, float(err[safe].max())
    assert safe.float().mean() > 0.5


def test_ste_and_mask_blend_and_no_adjoint_nodes():
    x = synthetic_frames(2, 32, 48, seed=7).cuda()
    x[0, :, :4] = 1.3          # outside [0, 1]: JPEG clamps first, the gradient is masked there (valuemetric.py:41)
    xg = x.clone().requires_grad_(True)
    y, _ = G.JPEG(40, 80)(xg, None, 50)
    assert y.requires_grad
    dy = _rand(2, 3, 32, 48, seed=8)
    y.backward(dy)
    inside = (x >= 0) & (x <= 1)
    assert torch.equal(xg.grad[inside], dy[inside]) and (xg.grad[~inside] == 0).all()
    xg = x.clamp(0, 1).requires_grad_(True)
    y, _ = G.MedianFilter(3, 3)(xg, None, 3)
    y.backward(dy)
    assert torch.equal(xg.grad, dy)
    # mask blend
    a, b = _rand(2, 3, 8, 9, seed=9).requires_grad_(True), _rand(2, 3, 8, 9, seed=10).requires_grad_(True)
    m = (torch.rand(2, 1, 8, 9, generator=torch.Generator().manual_seed(1)) > 0.4).float().cuda()
    out = G.mask_blend(a, b, m)
    d = _rand(2, 3, 8, 9, seed=11)
    out.backward(d)
    assert torch.equal(a.grad, d * m) and torch.equal(b.grad, d * (1 - m))
    # an op without adjoint kernel raises in backward instead of cutting the graph
    xg = x.clamp(0, 1).requires_grad_(True)
    mf = G.MedianFilter(3, 3)
    mf.passthrough = False
    y, _ = mf(xg, None, 3)
    with pytest.raises(NotImplementedError, match="MedianFilter"):
        y.sum().backward()


@pytest.mark.parametrize("k,H,W", [(3, 20, 24), (17, 40, 33), (9, 9, 30), (5, 5, 5)])
def test_gaussian_blur_adjoint(k, H, W):
    """valuemetric.py:108-128 (torchvision gaussian_blur, reflection padding): vs_gaussian_blur_bwd against autograd of the oracle's conv form"""
    x0 = synthetic_frames(2, H, W, seed=15)
    xc = x0.clone().requires_grad_(True)
    y = OA.gaussian_blur(xc, k)
    dy = _rand(2, 3, H, W, seed=16)
    y.backward(dy.cpu())
    xg = x0.cuda().requires_grad_(True)
    yg = G.gaussian_blur(xg, k)
    assert (yg.detach().cpu() - y.detach()).abs().max() < 1e-5
    yg.backward(dy)
    assert (xg.grad.cpu() - xc.grad).abs().max() <= 2e-5 * xc.grad.abs().max()


@pytest.mark.parametrize("factor", [0.07, -0.1, 0.31, -0.45])
def test_hue_adjoint(factor):
    """valuemetric.py:157-171 (torchvision adjust_hue): the hand-written chain rule through the RGB -> HSV -> RGB round trip against autograd of
    the oracle's restatement.  The map is piecewise linear in RGB with jumps where the hue sector changes: pixels whose outputs move by more than
    1e-3 under a 1e-6 input perturbation (a sector boundary within rounding) are left out of the comparison."""
    x0 = synthetic_frames(3, 40, 56, seed=17, kind="uniform")
    xc = x0.clone().requires_grad_(True)
    y = OA.hue(xc, factor)
    dy = _rand(3, 3, 40, 56, seed=18)
    y.backward(dy.cpu())
    xg = x0.cuda().requires_grad_(True)
    yg = G.color_op(xg, "hue", factor)
    assert (yg.detach().cpu() - y.detach()).abs().max() < 1e-5
    yg.backward(dy)
    with torch.no_grad():
        near = ((OA.hue(x0 + 1e-6, factor) - y).abs() > 1e-3).any(1, keepdim=True) | ((OA.hue(x0 - 1e-6, factor) - y).abs() > 1e-3).any(1, keepdim=True)
    safe = (~near).expand_as(y)
    err = (xg.grad.cpu() - xc.grad).abs()
    assert safe.float().mean() > 0.9
    assert err[safe].max() <= 5e-4 * xc.grad.abs().max(), float(err[safe].max())


@pytest.mark.parametrize("angle,expand,H,W", [(7, False, 40, 56), (-10, False, 33, 33), (90, True, 24, 40), (-90, True, 31, 17), (0, False, 16, 16)])
def test_rotate_adjoint(angle, expand, H, W):
    """geometric.py:28-59 (torchvision rotate, NEAREST): gather-form adjoint against autograd of grid_sample(mode='nearest')"""
    x0 = synthetic_frames(2, H, W, seed=19)
    xc = x0.clone().requires_grad_(True)
    y = OA.rotate(xc, angle, expand=expand)
    dy = _rand(*y.shape, seed=20)
    y.backward(dy.cpu())
    xg = x0.cuda().requires_grad_(True)
    yg = G.rotate(xg, angle, expand=expand)
    assert yg.shape == y.shape and (yg.detach().cpu() - y.detach()).abs().max() < 1e-6
    yg.backward(dy)
    assert (xg.grad.cpu() - xc.grad).abs().max() <= 1e-5 * xc.grad.abs().max().clamp_min(1e-6)


@pytest.mark.parametrize("scale,H,W", [(0.1, 40, 56), (0.5, 48, 48), (0.3, 33, 70)])
def test_perspective_adjoint(scale, H, W):
    """geometric.py:127-183 (torchvision perspective, BILINEAR, zero fill): gather-form adjoint against autograd of grid_sample"""
    torch.manual_seed(int(scale * 100))
    sp, ep = G.Perspective.get_perspective_params(W, H, scale)
    x0 = synthetic_frames(2, H, W, seed=21)
    xc = x0.clone().requires_grad_(True)
    y = OA.perspective(xc, sp, ep)
    dy = _rand(2, 3, H, W, seed=22)
    y.backward(dy.cpu())
    xg = x0.cuda().requires_grad_(True)
    yg = G.perspective(xg, sp, ep)
    assert (yg.detach().cpu() - y.detach()).abs().max() < 1e-4
    yg.backward(dy)
    assert (xg.grad.cpu() - xc.grad).abs().max() <= 1e-3 * xc.grad.abs().max()


@pytest.mark.parametrize("kind", ["mse", "yuv"])
def test_perceptual_and_decoding_loss_nodes(kind):
    imgs = synthetic_frames(3, 40, 48, seed=12)
    iw = (imgs + 0.01 * torch.randn(imgs.shape, generator=torch.Generator().manual_seed(2))).requires_grad_(True)
    ref = OL.perceptual(kind, imgs, iw)
    (ref * 3.0).backward()
    iwg = iw.detach().cuda().requires_grad_(True)
    got = AG.percep_loss(imgs.cuda(), iwg, kind)
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * max(float(ref.detach()), 1e-9) + 1e-12
    (got * 3.0).backward()
    assert (iwg.grad.cpu() - iw.grad).abs().max() <= 1e-5 * float(iw.grad.abs().max())
    preds = _rand(5, 17, seed=13).requires_grad_(True)
    msgs = synthetic_msgs(5, 16, seed=14)
    lp = preds.detach().cpu().requires_grad_(True)
    ref = OL.decoding_loss(lp / 2.0, msgs, None)
    ref.backward()
    got = AG.decoding_loss(preds, msgs, temperature=2.0)
    got.backward()
    assert abs(float(got) - float(ref)) < 1e-6 and (preds.grad.cpu() - lp.grad).abs().max() < 1e-7


# ------------------------------------------------------------------------------------------------ the shell's adjoint
@pytest.mark.parametrize("H,W,step,mode,lowres,nf", [(72, 88, 1, "repeat", False, 3), (64, 64, 1, "repeat", False, 2), (80, 72, 2, "repeat", False, 5),
                                                     (80, 72, 2, "repeat", True, 6), (70, 90, 3, "alternate", False, 7),
                                                     (70, 90, 2, "interpolate", True, 7)])
def test_embed_tail_adjoint(H, W, step, mode, lowres, nf):
    """d(delta) of: key-frame expansion -> (x low-res heat-map) -> resize up -> blend -> attenuation(imgs, imgs_w) -> clamp, against autograd
    through the oracle's operators"""
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    model = make_model(spec, sd)
    eng = model._engine()
    S = spec.img_size
    imgs = synthetic_frames(nf, H, W, seed=20)
    nkey = (nf + step - 1) // step
    delta = (torch.rand(nkey, 1, S, S, generator=torch.Generator().manual_seed(3)) * 2 - 1).requires_grad_(True)
    # oracle
    pw = R.apply_video_mode(delta, nf, step, mode)
    if lowres:
        pw = R.jnd_heatmaps(sd, spec, F.interpolate(imgs, size=(S, S), **AA)) * pw
    if (H, W) != (S, S):
        pw = F.interpolate(pw, size=(H, W), **AA)
    iw = spec.scaling_i * imgs + spec.scaling_w * pw
    if not lowres:
        iw = imgs + R.jnd_heatmaps(sd, spec, imgs) * (iw - imgs)
    iw = torch.clamp(iw, 0, 1)
    d_out = torch.randn(iw.shape, generator=torch.Generator().manual_seed(4))
    iw.backward(d_out)
    # HIP: forward tail, then the three adjoint kernels
    x = imgs.cuda()
    out, preds_w = torch.empty_like(x), torch.empty(nf, 1, H, W, device="cuda")
    hm_low = None
    if lowres:
        rgb, _ = eng.resize_pre(x, (S, S), True, want_rgb=True, tag="t.rs")
        hm_low = eng.jnd_lowres(rgb).clone()
    vm = N.VIDEO_MODES[mode]
    eng.embed_tail(x, out, delta.detach().cuda().contiguous(), step=step, video_mode=vm, hmap_low=hm_low, attenuate=(1 if lowres else 2), clamp=True,
                   antialias=True, scaling_i=spec.scaling_i, scaling_w=spec.scaling_w, preds_w=preds_w)
    assert (out.cpu() - iw.detach()).abs().max() < 2e-6
    L, st = eng.lib, N.stream()
    hm_full = None if lowres else eng.jnd_full(x)
    g_full = torch.empty(nf, 1, H, W, device="cuda")
    dog = d_out.cuda()
    N.check(L.vs_embed_tail_bwd(N.ptr(x), N.ptr(preds_w), N.ptr(hm_full), N.ptr(dog), None, nf, H, W, 1, 1, spec.scaling_i, spec.scaling_w, N.ptr(g_full),
                                st), "vs_embed_tail_bwd")
    g_low = AG.resize_bwd(g_full, (S, S), True) if (H, W) != (S, S) else g_full
    dd = torch.empty(nkey, 1, S, S, device="cuda")
    N.check(L.vs_tail_key_reduce(N.ptr(g_low), N.ptr(hm_low), nf, 1, S, S, step, vm, nkey, N.ptr(dd), st), "vs_tail_key_reduce")
    ref = delta.grad
    assert (dd.cpu() - ref).abs().max() <= 2e-5 * float(ref.abs().max()), float((dd.cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("arch", ["tiny", "tiny_chunky"])
def test_detector_input_gradient_matches_oracle_autograd(arch):
    """DetectTrainFn: d loss / d imgs_aug (the path of the decoding loss back to the embedder) and the parameter gradients in one pass.
    tiny_chunky: ChunkySeal's overlapping 4 x 4 stride-2 stem and odd feature maps (31 -> 15 -> 7 -> 3: the 2 x 2 stride-2 convs drop the last
    row / column, whose pixels receive no gradient), channel counts that are not multiples of 4"""
    spec = tiny_spec() if arch == "tiny" else tiny_spec(yuv=False, in_ch=3, out_ch=3, dims=[18, 36, 54, 90], stem_stride=2, hidden=32, nbits=16)
    sd = make_state_dict(spec, seed=3 if arch == "tiny" else 4)
    S = spec.img_size
    x = synthetic_frames(3, S, S, seed=30)
    names = [k for k in sd if k.startswith("detector.") and sd[k].dtype.is_floating_point]
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sdg[k].requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    dl = torch.randn(3, spec.nbits + 1, generator=torch.Generator().manual_seed(5))
    R.extractor_forward(sdg, spec, xc).backward(dl)
    model = make_model(spec, sd).train()
    xg = x.cuda().requires_grad_(True)
    named = AG._named_unique(model.detector, "detector.")
    preds = AG.DetectTrainFn.apply(model, xg, [k for k, _ in named], *[p for _, p in named])
    preds.backward(dl.cuda())
    assert (xg.grad.cpu() - xc.grad).abs().max() <= 2e-4 * float(xc.grad.abs().max())
    for k, p in named:
        rf = sdg[k].grad
        assert (p.grad.cpu() - rf).abs().max() <= 3e-4 * float(rf.abs().max()) + 1e-9, k
    # data-only pass (what the adaptive-weight probes use): same input gradient, no parameter gradient touched
    model.zero_grad()
    xg2 = x.cuda().requires_grad_(True)
    preds = AG.DetectTrainFn.apply(model, xg2, [k for k, _ in named], *[p for _, p in named])
    (gx,) = torch.autograd.grad(preds, xg2, dl.cuda())
    assert torch.equal(gx, xg.grad) and all(p.grad is None for _, p in named)
    # a second forward invalidates the first graph's operands: loud
    p1 = AG.DetectTrainFn.apply(model, xg2, [k for k, _ in named], *[p for _, p in named])
    AG.DetectTrainFn.apply(model, xg2, [k for k, _ in named], *[p for _, p in named])
    with pytest.raises(RuntimeError, match="stale"):
        p1.sum().backward()


# ------------------------------------------------------------------------------------------------ train.py's inner loop against the reference
def _setup(spec, sd, meta):
    model = make_model(spec, sd)
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs=meta["augs"], augs_params=meta["augs_params"], num_augs=meta["num_augs"])
    model.train()
    if meta["is_video"]:
        model.step_size = meta["step"]
    imgs = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"], kind=meta["kind"])
    msgs = synthetic_msgs(1 if meta["is_video"] else meta["n"], spec.nbits, seed=meta["seed"])
    masks = torch.ones(meta["n"], 1, meta["h"], meta["w"])
    return model, imgs, msgs, masks


def _summaries(named_grads):
    rows = {}
    for k, g in named_grads:
        gd = g.detach().double().flatten().cpu()
        rows[k] = np.array([float(gd.norm()), float(gd.sum()), float((gd * projection_vector(k, gd.numel())).sum())])
    return rows


def _check_against_fixture(g, grads, what, unet_tol):
    """3e-3 of the tensor's own gradient norm (scaled like tests/test_gpu_bwd.py) for every tensor of the tiny architecture and for the
    extractor of VideoSeal 1.0; its U-Net (12 M ReLU decisions per frame) gets the ReLU-flip bound (measured: 9.5e-3; the oracle's own
    fp32-vs-fp64 spread is 1.3e-2) -- the shared-decision test below holds the same tensors to 2e-3 element-wise"""
    names = [str(k) for k in g["grad_names"]]
    ref = g["grad_summary"]
    gmax = ref[:, 0].max()
    got = _summaries([(k, grads[k]) for k in names])
    worst = {"detector": 0.0, "embedder": 0.0}
    for i, k in enumerate(names):
        fam = "detector" if k.startswith("detector.") else "embedder"
        scale = max(ref[i, 0], 1e-4 * gmax) * max(1.0, np.sqrt(grads[k].numel()) / 16)
        dev = float(np.abs(got[k] - ref[i]).max() / scale)
        worst[fam] = max(worst[fam], dev)
        tol = 3e-3 if fam == "detector" else unet_tol
        assert dev <= tol, (what, k, got[k], ref[i], dev)
        nerr = abs(got[k][0] - ref[i, 0]) / max(ref[i, 0], 1e-4 * gmax)
        assert nerr <= tol, (what, k, "norm", got[k][0], ref[i, 0])
    print(f"{what}: worst summary deviation detector {worst['detector']:.2e}, U-Net {worst['embedder']:.2e}")


def _spec_sd(name):
    if name.startswith("vs10"):
        spec = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
        return spec, make_state_dict(spec, seed=0)
    if name.startswith("tinyv"):          # the legacy family; position tables seeded as in tests/golden/make_golden_bwd.py::legacy_sd
        from oracle.weights import legacy_tiny_spec
        spec = legacy_tiny_spec()
        sd = make_state_dict(spec, seed=6)
        gen = torch.Generator().manual_seed(9)
        for k in sd:
            if k.endswith(("pos_embed", "rel_pos_h", "rel_pos_w")):
                sd[k] = 0.2 * torch.randn(sd[k].shape, generator=gen)
        return spec, sd
    spec = tiny_spec()
    return spec, make_state_dict(spec, seed=3)


CASES = ["tiny_bwd_img_recipe", "tiny_bwd_img_balanced", "tiny_bwd_vid_recipe", "vs10_bwd_img_recipe", "tinyv_bwd_img_balanced"]


@pytest.mark.parametrize("name", CASES)
def test_unmodified_train_loop_reproduces_the_reference_gradients(name):
    """train.py:626-643 verbatim on the HIP module; the criterion is the restated VideosealLoss (oracle/loss.py, pinned to the reference's class)
    evaluated with torch on the graph-carrying outputs, exactly what the reference's own loss object would do"""
    spec, sd = _spec_sd(name)
    g = load_golden(name)
    meta = g["meta"]
    model, imgs, msgs, masks = _setup(spec, sd, meta)
    torch.manual_seed(meta["torch_seed"])
    imgs_d = imgs.cuda()
    outputs = model(imgs_d, masks.cuda(), msgs, is_video=meta["is_video"])                  # train.py:627
    assert outputs["selected_aug"] == meta["selected_aug"]
    outputs["preds"] /= meta["temperature"]                                                  # train.py:628
    last_layer = model.embedder.get_last_layer()                                             # train.py:631
    loss, log = OL.videoseal_loss(imgs_d, outputs["imgs_w"], outputs["masks"], outputs["msgs"].cuda(), outputs["preds"], last_layer=last_layer,
                                  **meta["loss_kw"])
    (loss / meta["accumulation"]).backward()                                                 # train.py:641-643
    assert (outputs["preds"].detach().cpu() - torch.from_numpy(g["preds"])).abs().max() < 1e-3
    for k, v in meta["log"].items():
        tol = (3e-3 if k.startswith("scale_") else 2e-4) * max(1.0, abs(v))
        assert abs(float(log[k]) - v) <= tol, (k, float(log[k]), v)
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(str(k) for k in g["grad_names"]) <= set(grads), sorted(set(str(k) for k in g["grad_names"]) - set(grads))[:5]
    unet_tol = 3e-2 if name.startswith("vs10") else 3e-3
    _check_against_fixture(g, grads, name, unet_tol)
    for k in [str(k) for k in g["full_names"]]:
        rf = torch.from_numpy(g["grad." + k])
        tol = 3e-3 if k.startswith("detector.") else unet_tol
        assert (grads[k].cpu() - rf).abs().max() <= tol * rf.abs().max() + 1e-9, k


@pytest.mark.parametrize("name", ["tiny_bwd_img_balanced", "tiny_bwd_vid_recipe", "vs10_bwd_img_recipe"])
def test_all_gradients_match_oracle_autograd_with_shared_relu_decisions(name):
    """the same step against autograd through the oracle run with the HIP forward's ReLU decisions: all 161 / 339 tensors, tight"""
    spec, sd = _spec_sd(name)
    g = load_golden(name)
    meta = g["meta"]
    names = [str(k) for k in g["grad_names"]]
    model, imgs, msgs, masks = _setup(spec, sd, meta)
    torch.manual_seed(meta["torch_seed"])
    step = GeneratorStep(model, percep_loss=meta["loss_kw"]["percep_loss"], percep_weight=meta["loss_kw"]["percep_weight"],
                         decode_weight=meta["loss_kw"]["decode_weight"], balanced=meta["loss_kw"]["balanced"], temperature=meta["temperature"])
    _, log, outputs = step.step(imgs.cuda(), masks.cuda(), msgs, is_video=meta["is_video"], accumulation_steps=meta["accumulation"])
    hip = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    masks_relu = hip_relu_masks(model, model._last_train_saved)
    # oracle with the same ReLU decisions
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sdg[k].requires_grad_(True)
    aug = OA.Augmenter(meta["augs"], meta["augs_params"], meta["num_augs"])
    torch.manual_seed(meta["torch_seed"])
    real_F = R.F
    try:
        R.F = _MaskedF(masks_relu)
        if meta["is_video"]:
            out = R.forward_video(sdg, spec, imgs, masks, msgs, aug, bn={}, step_size=meta["step"])
        else:
            out = R.forward_image(sdg, spec, imgs, masks, msgs, aug, bn={})
        assert R.F.i == len(masks_relu)
    finally:
        R.F = real_F
    assert out["selected_aug"] == outputs["selected_aug"]
    total, olog = OL.videoseal_loss(imgs, out["imgs_w"], out["masks"], out["msgs"], out["preds"] / meta["temperature"],
                                    last_layer=sdg[meta["last_layer"]], **meta["loss_kw"])
    (total / meta["accumulation"]).backward()
    for k in olog:
        assert abs(float(log[k]) - float(olog[k])) <= 2e-4 * max(1.0, abs(float(olog[k]))), (k, float(log[k]), float(olog[k]))
    errs = []
    for k in names:
        rf = sdg[k].grad
        errs.append((float((hip[k] - rf).abs().max() / rf.abs().max().clamp_min(1e-12)), k))
    errs.sort(reverse=True)
    print(f"{name}: worst element-wise relative gradient errors {[(round(e, 6), k) for e, k in errs[:4]]}, median {errs[len(errs) // 2][0]:.2e}")
    assert errs[0][0] < 2e-3, errs[:6]


def test_two_optimizer_steps_follow_the_oracle():
    """two SGD steps of the restated train_one_epoch inner loop (zero_grad -> forward -> loss -> backward -> step): the detector's weights (smooth
    network) against autograd through the oracle, tightly; the U-Net's within the ReLU-flip bound"""
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    g = load_golden("tiny_bwd_img_recipe")
    meta = g["meta"]
    model, imgs, msgs, masks = _setup(spec, sd, meta)
    lr = 0.05
    params = list(model.embedder.parameters()) + list(model.detector.parameters())            # train.py:330
    opt = torch.optim.SGD(params, lr=lr)
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    osd = {k: v.clone() for k, v in sd.items()}
    for it in range(2):
        torch.manual_seed(100 + it)
        opt.zero_grad()
        out = model(imgs.cuda(), masks.cuda(), msgs, is_video=False)
        loss, _ = OL.videoseal_loss(imgs.cuda(), out["imgs_w"], out["masks"], out["msgs"].cuda(), out["preds"],
                                    last_layer=model.embedder.get_last_layer(), **meta["loss_kw"])
        loss.backward()
        opt.step()
        # oracle step on its own copy of the weights (BatchNorm running statistics follow too)
        sdg = {k: v.clone() for k, v in osd.items()}
        for k in names:
            sdg[k].requires_grad_(True)
        bn = {}
        torch.manual_seed(100 + it)
        oo = R.forward_image(sdg, spec, imgs, masks, msgs, OA.Augmenter(meta["augs"], meta["augs_params"], meta["num_augs"]), bn=bn)
        ol, _ = OL.videoseal_loss(imgs, oo["imgs_w"], oo["masks"], oo["msgs"], oo["preds"], last_layer=sdg[meta["last_layer"]], **meta["loss_kw"])
        ol.backward()
        assert oo["selected_aug"] == out["selected_aug"]
        assert abs(float(loss) - float(ol)) < 2e-4
        for k in names:
            osd[k] = (sdg[k] - lr * sdg[k].grad).detach()
        osd.update({k: v.detach() for k, v in bn.items()})
    new = {k: v.cpu() for k, v in model.state_dict().items()}
    for k in names:
        moved = (osd[k] - sd[k]).abs().max()
        err = (new[k] - osd[k]).abs().max()
        tol = (1e-2 if k.startswith("detector.") else 1e-1) * float(moved) + 1e-7
        assert err <= tol, (k, float(err), float(moved))
    for k in osd:
        if "running_" in k:
            assert (new[k] - osd[k]).abs().max() < 1e-4, k
        if "num_batches_tracked" in k:
            assert int(new[k]) == int(osd[k]) == int(sd[k]) + 2


def test_frozen_embedder_and_eval_batchnorm():
    """train.py:507-523: embedder.requires_grad_(False) -> only the detector receives gradients; a trainable embedder in eval() mode
    (BatchNorm on its running statistics) still has a backward"""
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    g = load_golden("tiny_bwd_img_recipe")
    meta = g["meta"]
    model, imgs, msgs, masks = _setup(spec, sd, meta)
    model.embedder.requires_grad_(False)
    torch.manual_seed(meta["torch_seed"])
    out = model(imgs.cuda(), masks.cuda(), msgs, is_video=False)
    assert not out["imgs_w"].requires_grad and out["preds"].requires_grad
    OL.decoding_loss(out["preds"], out["msgs"].cuda(), None).backward()
    assert all(p.grad is None for p in model.embedder.parameters()) and all(p.grad is not None for p in model.detector.parameters())
    # eval-mode BatchNorm with gradients: against the oracle's autograd with bn=None (running statistics), shared ReLU decisions
    model2, imgs, msgs, masks = _setup(spec, sd, meta)
    model2.embedder.eval()
    torch.manual_seed(meta["torch_seed"])
    out = model2(imgs.cuda(), masks.cuda(), msgs, is_video=False)
    OL.decoding_loss(out["preds"], out["msgs"].cuda(), None).backward()
    names = [k for k, p in model2.named_parameters() if k.startswith("embedder.") and p.grad is not None]
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sdg[k].requires_grad_(True)
    torch.manual_seed(meta["torch_seed"])
    real_F = R.F
    try:
        R.F = _MaskedF(hip_relu_masks(model2, model2._last_train_saved))
        oo = R.forward_image(sdg, spec, imgs, masks, msgs, OA.Augmenter(meta["augs"], meta["augs_params"], meta["num_augs"]), bn=None)
    finally:
        R.F = real_F
    OL.decoding_loss(oo["preds"], oo["msgs"], None).backward()
    hip = dict(model2.named_parameters())
    worst = max(float((hip[k].grad.cpu() - sdg[k].grad).abs().max() / sdg[k].grad.abs().max().clamp_min(1e-12)) for k in names)
    assert worst < 2e-3, worst
    assert torch.equal(model2.state_dict()["embedder.unet.inc.double_conv.1.num_batches_tracked"].cpu(), sd["embedder.unet.inc.double_conv.1.num_batches_tracked"])


def test_legacy_card_detector_fine_tuning_under_the_unmodified_loop():
    """The legacy architecture (RMSNorm / SiLU U-Net, ViT extractor) with the embedder frozen (train.py:507-523): `model(imgs, masks, msgs)` ->
    decoding loss -> `loss.backward()` fills the gradient of every ViT parameter (attention incl. the relative-position tables, MLP, neck,
    pixel decoder) through the HIP backward; same values as `DetectorStep.step` on the forward's own `imgs_aug` (which tests/test_gpu_bwd.py
    holds to oracle autograd)."""
    from oracle.weights import legacy_tiny_spec
    from videoseal_amd.training import DetectorStep
    spec = legacy_tiny_spec()
    sd = make_state_dict(spec, seed=5)
    model = make_model(spec, sd)
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs={"identity": 1}, augs_params={}, num_augs=1)
    model.train()
    model.embedder.requires_grad_(False)
    imgs = synthetic_frames(3, 64, 64, seed=12).cuda()
    msgs = synthetic_msgs(3, spec.nbits, seed=12)
    masks = torch.ones(3, 1, 64, 64, device="cuda")
    out = model(imgs, masks, msgs, is_video=False)
    assert out["preds"].requires_grad and not out["imgs_w"].requires_grad
    OL.decoding_loss(out["preds"], out["msgs"].cuda(), None).backward()
    got = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    assert all(k.startswith("detector.") for k in got) and len(got) == sum(1 for k, _ in model.detector.named_parameters())
    _, _, ref = DetectorStep(model).step(out["imgs_aug"].detach(), msgs, accumulate=False)
    torch.cuda.synchronize()
    for k, v in got.items():
        r = ref[k].reshape(v.shape)
        assert (v - r).abs().max() <= 1e-6 * r.abs().max() + 1e-12, k


def test_legacy_card_full_training_step_matches_oracle_autograd():
    """The whole legacy architecture trainable under the unmodified loop: RMSNorm / SiLU U-Net (common.py:172-179; vs_rmsnorm_act_bwd, vs_act_bwd),
    RGB watermark without JND, ViT extractor.  Frames at 72 x 88 (resize both ways around the 64 x 64 networks), perceptual + decoding terms
    with fixed weights: every gradient against torch autograd through the oracle (CPU, fp32).  All activations are smooth here (no ReLU
    decisions to share): the comparison is element-wise at 2e-3 of each tensor's largest entry."""
    from oracle.weights import legacy_tiny_spec
    spec = legacy_tiny_spec()
    sd = make_state_dict(spec, seed=5)
    gen = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith(("pos_embed", "rel_pos_h", "rel_pos_w")):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=gen)
    model = make_model(spec, sd)
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs={"identity": 1}, augs_params={}, num_augs=1)
    model.train()
    imgs = synthetic_frames(3, 72, 88, seed=13)
    msgs = synthetic_msgs(3, spec.nbits, seed=13)
    masks = torch.ones(3, 1, 72, 88)
    kw = dict(percep_loss="mse", percep_weight=1.0, detect_weight=0.0, decode_weight=0.5, balanced=False)
    out = model(imgs.cuda(), masks.cuda(), msgs, is_video=False)
    assert out["imgs_w"].requires_grad and out["preds"].requires_grad
    loss, _ = OL.videoseal_loss(imgs.cuda(), out["imgs_w"], out["masks"], out["msgs"].cuda(), out["preds"], last_layer=None, **kw)
    loss.backward()
    torch.cuda.synchronize()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    got = dict(model.named_parameters())
    sdg = {k: v.clone() for k, v in sd.items()}
    for k in names:
        sdg[AG._ALIASES.get(k, k)].requires_grad_(True)
    oo = R.forward_image(sdg, spec, imgs, masks, msgs, OA.Augmenter({"identity": 1}, {}, 1))
    lref, _ = OL.videoseal_loss(imgs, oo["imgs_w"], oo["masks"], oo["msgs"], oo["preds"], last_layer=None, **kw)
    lref.backward()
    assert abs(float(loss) - float(lref)) <= 1e-5 * abs(float(lref))
    worst, n = 0.0, 0
    for k in names:
        ref = sdg[AG._ALIASES.get(k, k)].grad
        if ref is None:
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k
            continue
        assert got[k].grad is not None, k
        err = float((got[k].grad.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        worst, n = max(worst, err), n + 1
        assert err < 2e-3, (k, err)
    assert n > 100
    print(f"legacy card: worst relative gradient error over {n} tensors: {worst:.2e}")


ALL_AUGS = {   # configs/all_augs.yaml of the reference (train.py's default --augmentation_config), restated: names and parameter ranges
    "identity": {}, "jpeg": dict(min_quality=40, max_quality=80), "resize": dict(min_size=0.7, max_size=1.5), "crop": dict(min_size=0.5, max_size=1.0),
    "rotate": dict(min_angle=-10, max_angle=10, do90=True), "hflip": {}, "perspective": dict(min_distortion_scale=0.1, max_distortion_scale=0.5),
    "gaussian_blur": dict(min_kernel_size=3, max_kernel_size=17), "median_filter": dict(min_kernel_size=3, max_kernel_size=3),
    "brightness": dict(min_factor=0.5, max_factor=2), "contrast": dict(min_factor=0.5, max_factor=2.0), "saturation": dict(min_factor=0.5, max_factor=2),
    "hue": dict(min_factor=-0.1, max_factor=0.1), "h264": dict(min_crf=28, max_crf=36), "h264rgb": dict(min_crf=28, max_crf=36),
    "h265": dict(min_crf=28, max_crf=36)}


@pytest.mark.parametrize("name", sorted(ALL_AUGS))
def test_every_augmentation_of_the_default_training_config_has_a_backward(name):
    """train.py's default augmentation set (configs/all_augs.yaml: 16 entries) inside the unmodified loop: whichever augmentation the augmenter
    draws, `loss.backward()` reaches the embedder -- exact adjoints for the geometric / colour / blur ops, the reference's own straight-through
    estimators for JPEG / MedianFilter / the codecs.  Every parameter of both networks gets a finite gradient, the embedder's is not zero."""
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    model = make_model(spec, sd)
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs={name: 1}, augs_params={k: v for k, v in ALL_AUGS.items() if v}, num_augs=1)
    model.train()
    video = name in ("h264", "h264rgb", "h265")
    n = 4
    imgs = synthetic_frames(n, 72, 88, seed=23).cuda()
    msgs = synthetic_msgs(1 if video else n, spec.nbits, seed=23)
    masks = torch.ones(n, 1, 72, 88, device="cuda")
    torch.manual_seed(5)
    if video:
        model.step_size = 2
    out = model(imgs, masks, msgs, is_video=video)
    assert out["imgs_aug"].requires_grad and out["preds"].requires_grad
    loss, _ = OL.videoseal_loss(imgs, out["imgs_w"], out["masks"], out["msgs"].cuda(), out["preds"], last_layer=None,
                                percep_loss="mse", percep_weight=0.0, detect_weight=0.0, decode_weight=1.0, balanced=False)
    loss.backward()
    torch.cuda.synchronize()
    ge = [p.grad for k, p in model.named_parameters() if k.startswith("embedder.") and p.requires_grad]
    gd = [p.grad for k, p in model.named_parameters() if k.startswith("detector.")]
    assert all(g is not None and torch.isfinite(g).all() for g in ge + gd)
    assert max(float(g.abs().max()) for g in ge) > 0.0, f"{out['selected_aug']}: no gradient reached the embedder"


def test_temporal_augmentation_adjoints():
    """whole-frame gathers (DropFrame / SpeedChange / TemporalReorder, video.py:283-405, 491-529) and WindowAveraging (video.py:411-486): exact
    adjoints against torch autograd of the same index / window arithmetic; the augmentation classes run under autograd"""
    import random
    x0 = synthetic_frames(9, 24, 40, seed=31).cuda()
    idx = [0, 0, 2, 5, 5, 5, 8, 1, 3, 3, 7]
    xr = x0.clone().requires_grad_(True)
    dy = _rand(len(idx), 3, 24, 40, seed=32)
    xr[torch.tensor(idx, device="cuda")].backward(dy)
    xg = x0.clone().requires_grad_(True)
    yg = G.gather_frames(xg, idx)
    assert torch.equal(yg.detach(), x0[torch.tensor(idx, device="cuda")])
    yg.backward(dy)
    assert (xg.grad - xr.grad).abs().max() <= 1e-6 * xr.grad.abs().max()
    assert (xg.grad[4] == 0).all() and (xg.grad[6] == 0).all()           # frames nobody copied
    for ws, alpha in ((3, 0.4), (5, 0.7), (2, 1.0)):
        hw = ws // 2
        xr = x0.clone().requires_grad_(True)
        rows = []
        for i in range(9):
            a, b = max(0, i - hw), min(9, i + hw + 1)
            rows.append((1 - alpha) * xr[i] + alpha * xr[a:b].mean(0))
        yr = torch.stack(rows)
        dy = _rand(9, 3, 24, 40, seed=33)
        yr.backward(dy)
        xg = x0.clone().requires_grad_(True)
        yg = G.window_average(xg, ws, alpha)
        assert (yg.detach() - yr.detach()).abs().max() < 1e-6
        yg.backward(dy)
        assert (xg.grad - xr.grad).abs().max() <= 2e-6 * xr.grad.abs().max()
    random.seed(3)
    torch.manual_seed(3)
    for aug in (G.DropFrame(0.3), G.SpeedChange(0.5, 1.5) if hasattr(G, "SpeedChange") else None, G.TemporalReorder() if hasattr(G, "TemporalReorder") else None,
                G.WindowAveraging()):
        if aug is None:
            continue
        xg = x0.clone().requires_grad_(True)
        y, _ = aug(xg, None)
        y.sum().backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all() and abs(float(xg.grad.sum()) - y.numel()) <= 1e-3 * y.numel()
