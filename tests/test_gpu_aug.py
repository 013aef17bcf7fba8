"""Augmentation kernels vs the CPU oracle (torch / Pillow), and BASELINE config 3: clip -> embed -> fixed-strength chain
(JPEG -> Crop -> Resize -> colour ops) -> detect, compared with the oracle chain on the same watermarked frames."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment as A  # noqa: E402
from oracle import videoseal_ref as R  # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs  # noqa: E402
from oracle.weights import make_state_dict, tiny_spec  # noqa: E402
from tests._util import assert_decisions  # noqa: E402
from tests.test_gpu_e2e import make_model  # noqa: E402

CHAIN_MARGIN = 2e-5       # through the augmentation chain: measured logit error 2.2e-6 (profiles/r06a_decision_margins.txt)
from videoseal_amd import augmentation as G  # noqa: E402
from videoseal_amd import native as N  # noqa: E402


@pytest.fixture(scope="module")
def frames():
    return synthetic_frames(3, 93, 118, seed=31)


@pytest.mark.parametrize("hw,quality", [((64, 64), 60), ((93, 118), 40), ((50, 50), 90), ((37, 53), 75), ((256, 256), 50), ((768, 768), 80)])
def test_jpeg_is_bit_exact_with_pillow(hw, quality):
    x = synthetic_frames(2, hw[0], hw[1], seed=hw[0] + quality)
    x[0, :, :5, :5] = 1.7      # out-of-range values are clamped first (valuemetric.py:41)
    x[1, :, -3:, :] = -0.2
    ref = A.jpeg(x, quality)
    got, _ = G.JPEG()(x.cuda(), None, quality)
    assert torch.equal(got.cpu(), ref), "GPU JPEG differs from the Pillow/libjpeg round trip"


@pytest.mark.parametrize("hw,quality", [((768, 768), 40), ((64, 96), 75), ((256, 512), 90)])
def test_jpeg_vector_kernels_equal_the_scalar_ones(hw, quality):
    """round 6: frames made of whole 16 x 16 MCUs take the 16-byte-access kernels (4 x 2 pixels per thread in the colour / down-sampling stage,
    4 pixels per thread in the up-sampling / RGB stage, one launch for the blocks of all three components): the same bytes as the one-pixel-per-lane
    kernels (development switch 3) -- and as Pillow (test above: 64 x 64, 256 x 256 and 768 x 768 run the vector form, the other sizes the scalar one)"""
    x = synthetic_frames(3, hw[0], hw[1], seed=quality).cuda()
    x[0, :, :7, :9] = 1.3
    L = N.lib()
    got, _ = G.JPEG()(x, None, quality)
    L.vs_debug_set(3, 1)
    try:
        ref, _ = G.JPEG()(x, None, quality)
        torch.cuda.synchronize()
    finally:
        L.vs_debug_set(3, 0)
    assert torch.equal(got, ref)
    assert torch.equal(got.cpu(), A.jpeg(x.cpu(), quality))


@pytest.mark.parametrize("name,op,ref,vals", [
    ("brightness", G.Brightness, A.brightness, [0.1, 0.5, 1.5, 2.0]),
    ("contrast", G.Contrast, A.contrast, [0.1, 0.5, 1.5, 2.0]),
    ("saturation", G.Saturation, A.saturation, [0.0, 0.5, 1.5, 2.0]),
    ("hue", G.Hue, A.hue, [-0.4, -0.1, 0.1, 0.25, 0.5]),
])
def test_colour_ops(frames, name, op, ref, vals):
    for v in vals:
        got, m = op()(frames.cuda(), None, v)
        assert m is None
        assert (got.cpu() - ref(frames, v)).abs().max() < 2e-6, (name, v)      # fp32 pointwise, torchvision semantics restated


def test_grayscale_flip_crop_resize(frames):
    x = frames.cuda()
    mask = torch.rand(3, 1, 93, 118)
    assert (G.Grayscale()(x)[0].cpu() - A.grayscale(frames)).abs().max() < 1e-6
    im, mk = G.HorizontalFlip()(x, mask.cuda())
    assert torch.equal(im.cpu(), A.hflip(frames)) and torch.equal(mk.cpu(), A.hflip(mask))
    torch.manual_seed(7)
    im, mk = G.Crop()(x, mask.cuda(), 0.71)
    torch.manual_seed(7)
    th, tw = int(0.71 * 93), int(0.71 * 118)
    i = torch.randint(0, 93 - th + 1, size=(1,)).item(); j = torch.randint(0, 118 - tw + 1, size=(1,)).item()
    assert im.shape[-2:] == (th, tw) and torch.equal(im.cpu(), A.crop(frames, i, j, th, tw)) and torch.equal(mk.cpu(), A.crop(mask, i, j, th, tw))
    for s in (0.32, 0.71, 1.0, 1.4):
        im, mk = G.Resize()(x, mask.cuda(), s)
        size = (int(s * 93), int(s * 118))
        assert (im.cpu() - A.resize(frames, size)).abs().max() < 2e-6 and (mk.cpu() - A.resize(mask, size)).abs().max() < 2e-6


@pytest.mark.parametrize("angle", [5, 10, 30, 45, 90, -90, 100, -17])
def test_rotate(frames, angle):
    """geometric.py:28-59 through vs_aug_warp vs the restated torchvision grid + ATen grid_sample (nearest: a value is either equal or
    comes from the neighbouring pixel when the source coordinate sits within an ulp of a .5 boundary)."""
    x = frames.cuda()
    mask = (torch.rand(3, 1, 93, 118) > 0.5).float()
    im, mk = G.Rotate()(x, mask.cuda(), angle)
    base = angle // 90 * 90
    ref = A.rotate(A.rotate(frames, base, expand=True), angle - base)
    refm = A.rotate(A.rotate(mask, base, expand=True), angle - base)
    assert im.shape == ref.shape and mk.shape == refm.shape
    # (odd x even frames: a 90-degree turn puts source coordinates exactly on .5 ties, so even those are not a clean permutation)
    assert ((im.cpu() != ref).float().mean() < 2e-3) and ((mk.cpu() != refm).float().mean() < 2e-3)
    if angle == 90:          # even x even: exact pixel permutation
        ev = frames[..., :92, :].contiguous()
        assert torch.equal(G.Rotate()(ev.cuda(), None, 90)[0].cpu(), torch.rot90(ev, 1, dims=(-2, -1)))


@pytest.mark.parametrize("scale", [0.1, 0.3, 0.5, 0.8])
def test_perspective(frames, scale):
    """geometric.py:127-183: same corner draws as the reference (torch CPU RNG), bilinear grid_sample with zero fill."""
    x = frames.cuda()
    mask = torch.rand(3, 1, 93, 118)
    torch.manual_seed(3)
    im, mk = G.Perspective()(x, mask.cuda(), scale)
    torch.manual_seed(3)
    sp, ep = G.Perspective.get_perspective_params(118, 93, scale)
    ref, refm = A.perspective(frames, sp, ep), A.perspective(mask, sp, ep)
    # coordinates agree to a few ulp (summation order of the 3-term dot products): a bilinear sample moves by <= |grad| * 1e-4 px
    assert (im.cpu() - ref).abs().max() < 2e-4 and (mk.cpu() - refm).abs().max() < 5e-4
    assert (im.cpu() - ref).abs().mean() < 2e-6


@pytest.mark.parametrize("k", [3, 5, 9, 13, 17])
def test_gaussian_blur(frames, k):
    got, _ = G.GaussianBlur()(frames.cuda(), None, k)
    assert (got.cpu() - A.gaussian_blur(frames, k)).abs().max() < 2e-6


@pytest.mark.parametrize("k", [3, 5])
def test_median_filter(frames, k):
    got, _ = G.MedianFilter()(frames.cuda(), None, k)
    assert torch.equal(got.cpu(), A.median_filter(frames, k))      # selection only: exact


def test_augmenter_draws_strengths_in_the_reference_order():
    """augmenter.py:137-152 with value ops: the multinomial pick, then each op's own draw (valuemetric.py:27-31, 104-108), replayed
    here draw by draw.  (Names + crop draws against the REFERENCE Augmenter itself: tests/test_gpu_fwd.py, augmenter_picks.json.)"""
    augs = {"identity": 1, "crop": 1, "brightness": 1, "jpeg": 1, "hflip": 1}
    params = {"crop": {"min_size": 0.5, "max_size": 1.0}, "brightness": {"min_factor": 0.5, "max_factor": 2},
              "jpeg": {"min_quality": 40, "max_quality": 80}}
    aug = G.Augmenter(masks={"kind": "none"}, augs=augs, augs_params=params, num_augs=3).train()
    names = list(augs)
    cls = {"identity": "Identity", "crop": "Crop", "brightness": "Brightness", "jpeg": "JPEG", "hflip": "HorizontalFlip"}
    x = synthetic_frames(2, 64, 64, seed=3)
    for seed in range(8):
        torch.manual_seed(seed)
        out, mask, picked = aug(x.cuda(), x.cuda(), None, is_video=False, do_resize=True)
        torch.manual_seed(seed)
        ref, want = x, []
        for _ in range(3):
            n = names[torch.multinomial(torch.full((5,), 0.2), 1).item()]
            want.append(cls[n])
            if n == "crop":
                th = torch.randint(32, 65, size=(1,)).item(); tw = torch.randint(32, 65, size=(1,)).item()
                i, j = (0, 0) if (th, tw) == (64, 64) else (torch.randint(0, 64 - th + 1, size=(1,)).item(), torch.randint(0, 64 - tw + 1, size=(1,)).item())
                ref = A.resize(A.crop(ref, i, j, th, tw), (64, 64))
            elif n == "brightness":
                ref = A.brightness(ref, torch.rand(1).item() * 1.5 + 0.5)
            elif n == "jpeg":
                ref = A.jpeg(ref, torch.randint(40, 81, size=(1,)).item())
            elif n == "hflip":
                ref = A.hflip(ref)
        assert picked == "+".join(want), seed
        assert out.shape == x.shape and mask.shape == (2, 1, 64, 64)
        assert (out.cpu() - ref).abs().max() < (1e-5 if "JPEG" not in picked else 3e-2), (seed, picked)


@pytest.mark.parametrize("shape", [(3, 64, 64), (2, 70, 90), (1, 33, 47), (4, 144, 176)])
@pytest.mark.parametrize("crf", [0, 23, 28, 40, 51])
def test_h264_proxy_bit_exact_with_its_definition(shape, crf):
    """csrc/h264_proxy.hip vs oracle/h264_proxy.py (integer arithmetic: bit-exact), yuv420 and rgb planes.  The proxy stands in for
    the reference's libx264 round trip (augmentation/video.py:20-119), which cannot be pinned offline -- see the oracle header."""
    from oracle import h264_proxy as HP
    x = synthetic_frames(*shape, seed=9)
    for rgb in (False, True):
        got = G.h264_proxy(x.cuda(), crf, rgb_mode=rgb).cpu()
        ref = torch.from_numpy(HP.roundtrip(x.numpy(), crf, rgb))
        assert torch.equal(got, ref), (shape, crf, rgb, (got - ref).abs().max().item())


def test_h264_classes_follow_the_reference_interface():
    """video.py:147-205: crf drawn with torch.randint from [min_crf, max_crf]; odd sizes zero-padded to even; masks pass through;
    name2aug wiring; the pyav side path is loud when PyAV is missing"""
    x = synthetic_frames(5, 45, 63, seed=3).cuda()
    m = torch.ones(5, 1, 45, 63).cuda()
    torch.manual_seed(11)
    out, mo = G.H264(min_crf=28, max_crf=40)(x, m)
    torch.manual_seed(11)
    crf = torch.randint(28, 41, size=(1,)).item()
    xp = torch.nn.functional.pad(x, (0, 1, 0, 1))
    assert out.shape == (5, 3, 46, 64) and mo.shape == (5, 1, 46, 64)
    assert torch.equal(out, G.h264_proxy(xp, crf))
    assert torch.equal(G.H264rgb(20, 20)(xp)[0], G.h264_proxy(xp, 20, rgb_mode=True))
    assert torch.equal(G.H265(30, 30)(xp)[0], G.h264_proxy(xp, 30))
    assert torch.equal(G.VideoCompression(crf=34)(xp)[0], G.h264_proxy(xp, 34))
    assert G.name2aug["h264"] is G.H264 and G.name2aug["video_compression"] is G.VideoCompression
    with pytest.raises(ValueError):
        G.H264()(xp)
    vc = G.VideoCompression()
    vc.backend = "pyav"
    try:
        import av  # noqa: F401
    except ImportError:
        with pytest.raises(N.NativeError):
            vc(xp)
    # distortion grows with crf
    ps = [float(((G.h264_proxy(xp, q) - xp) ** 2).mean()) for q in (10, 23, 34, 46)]
    assert ps == sorted(ps)


def test_config3_clip_through_the_full_chain():
    """16-frame clip (BASELINE config 3, smaller frames for the CPU oracle) -> embed -> JPEG(60) -> Crop(0.71) -> Resize(0.8)
    -> Brightness(0.5) -> Contrast(1.5) -> Saturation(1.5) -> Hue(0.1) -> detect; vs the oracle chain on OUR watermarked frames."""
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    model = make_model(spec, sd)
    model.chunk_size, model.step_size, model.video_mode = 8, 2, "repeat"
    imgs = synthetic_frames(16, 144, 176, seed=77)
    msgs = synthetic_msgs(1, spec.nbits, seed=77)
    w = model.embed(imgs.cuda(), msgs, is_video=True)["imgs_w"]
    chain = G.Sequential(G.JPEG(), G.Crop(), G.Resize(), G.Brightness(), G.Contrast(), G.Saturation(), G.Hue())
    args = (60, 0.71, 0.8, 0.5, 1.5, 1.5, 0.1)
    torch.manual_seed(5)
    aug, _ = chain(w, None, args)
    # oracle chain on the same frames
    wc = w.cpu()
    torch.manual_seed(5)
    r = A.jpeg(wc, 60)
    th, tw = int(0.71 * 144), int(0.71 * 176)
    i = torch.randint(0, 144 - th + 1, size=(1,)).item(); j = torch.randint(0, 176 - tw + 1, size=(1,)).item()
    r = A.crop(r, i, j, th, tw)
    r = A.resize(r, (int(0.8 * th), int(0.8 * tw)))
    r = A.hue(A.saturation(A.contrast(A.brightness(r, 0.5), 1.5), 1.5), 0.1)
    assert aug.shape == r.shape
    assert (aug.cpu() - r).abs().max() < 1e-5
    preds = model.detect(aug, is_video=True)["preds"].cpu()
    pref = R.detect(sd, spec, r)["preds"]
    assert (preds - pref).abs().max() < 1e-3
    assert_decisions(preds, pref, margin=CHAIN_MARGIN, what="augmentation chain vs oracle", min_sure=0.99)
    acc = R.bit_accuracy(preds[:, 1:], msgs.expand(16, -1).float())
    acc_ref = R.bit_accuracy(pref[:, 1:], msgs.expand(16, -1).float())
    assert (acc - acc_ref).abs().max() < 1e-3


def test_temporal_reorder_and_window_averaging_match_the_reference_semantics():
    """augmentation/video.py:319-486 restated with torch on the CPU (chunk swap by python `random`; sliding-window mean blend)"""
    import random
    x = synthetic_frames(11, 24, 20, seed=31)
    xg = x.cuda()
    # TemporalReorder: same draws as the reference -> same permutation
    random.seed(5)
    y, _ = G.TemporalReorder()(xg, None, 3, 0.7)
    random.seed(5)
    nch = 11 // 3
    order = list(range(nch))
    for i in range(0, nch - 1, 2):
        if random.random() < 0.7 and i + 1 < nch:
            order[i], order[i + 1] = order[i + 1], order[i]
    ref = torch.cat([x[:9].view(nch, 3, *x.shape[1:])[order].reshape(-1, *x.shape[1:]), x[9:]], 0)
    assert torch.equal(y.cpu(), ref)
    short, _ = G.TemporalReorder()(xg[:5], None, 3, 1.0)
    assert short is xg[:5] or torch.equal(short, xg[:5])
    # WindowAveraging
    for ws, alpha in ((3, 1.0), (4, 0.4), (20, 0.6)):
        y, _ = G.WindowAveraging()(xg, None, ws, alpha)
        w = min(ws, 11)
        ref = x.clone()
        for i in range(11):
            a, b = max(0, i - w // 2), min(11, i + w // 2 + 1)
            ref[i] = (1 - alpha) * x[i] + alpha * torch.mean(x[a:b], dim=0, keepdim=True).squeeze(0)
        assert (y.cpu() - ref).abs().max() <= 1.2e-7, (ws, alpha)
    with pytest.raises(NotImplementedError, match="pyav"):
        G.VP9()(xg, None)


# ---- round 5: fused passes of the validation chains (vs_aug_color_chain, vs_aug_crop_resize_color, Sequential's fusion) ----------------
@pytest.mark.parametrize("ops", [
    [("brightness", 0.5), ("contrast", 1.5), ("saturation", 1.5), ("hue", 0.1)],          # configs[2]: two runs (cut in front of contrast)
    [("contrast", 0.7), ("contrast", 1.3), ("hue", -0.2)],                                # every contrast starts its own run
    [("saturation", 0.3), ("grayscale", 0.0), ("brightness", 1.7), ("hue", 0.45), ("saturation", 1.9), ("brightness", 0.9), ("hue", 0.1)],   # > 6 ops
    [("hue", 0.25)],
])
def test_fused_colour_chain_is_bit_identical_to_the_separate_ops(ops):
    x = synthetic_frames(3, 93, 118, seed=33).cuda()
    x[0, :, :4, :4] = 0.0                                      # gray pixels: the hue op's max == min branch
    ref = x
    for name, f in ops:
        ref = G.color_op(ref, name, f)
    got = G.color_chain(x, ops)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    for (name, f), fn in zip(ops[:1], [getattr(A, ops[0][0], None)]):          # and the first op against the oracle, as test_colour_ops does
        if fn is not None and name != "grayscale":
            assert (G.color_chain(x, ops[:1]).cpu() - fn(x.cpu(), f)).abs().max().item() < 2e-6


@pytest.mark.parametrize("H,W,crop,size,aa", [
    (768, 768, (111, 112, 545, 545), (386, 386), True),        # configs[2]: Crop(0.71) -> Resize(0.71)
    (93, 118, None, (47, 200), True),                          # no crop, down in y / up in x
    (93, 118, (0, 0, 93, 118), (150, 61), True),
    (64, 80, (5, 7, 33, 41), (33, 41), True),                  # identity-size resize of a crop
    (70, 66, (3, 2, 60, 60), (13, 17), True),                  # 4.6 x down: long taps
    (70, 66, (3, 2, 60, 60), (91, 77), False),                 # plain bilinear, up
    (40, 40, (30, 30, 10, 10), (64, 64), True),                # window at the corner
    # round 6 (row-streaming form): frame widths that are multiples of four (16-byte row pieces) with crop columns at every residue, the window's
    # right edge on the frame's, 3.3 x and 3.9 x down (the 128- and the 64-column tile), a strip taller than the weight table's 64 rows
    (96, 128, (5, 7, 64, 100), (40, 60), True),
    (96, 128, (1, 29, 95, 99), (120, 131), True),
    (256, 512, (0, 2, 256, 510), (77, 154), True),
    (400, 400, (3, 5, 390, 390), (100, 100), True),
    (64, 1024, (0, 0, 64, 1024), (200, 300), False),
])
@pytest.mark.parametrize("form", ["stream", "tile"])
def test_fused_crop_resize_colour_is_bit_identical_to_the_separate_launches(H, W, crop, size, aa, form):
    x = synthetic_frames(3, H, W, seed=H + W).cuda()
    ops = [("brightness", 0.5), ("saturation", 1.4), ("contrast", 1.5), ("hue", 0.1)]
    N.lib().vs_debug_set(4, 1 if form == "tile" else 0)          # development switch 4: the 32 x 8 tile kernel instead of the row-streaming one
    try:
        for use_ops in ([], ops):
            y = G.crop_flip(x, *crop) if crop is not None else x
            ref = G.resize(y, size, aa)
            for name, f in use_ops:
                ref = G.color_op(ref, name, f)
            got = G.crop_resize_color(x, crop, size, use_ops, antialias=aa)
            torch.cuda.synchronize()
            assert got.shape == ref.shape and torch.equal(got, ref), float((got - ref).abs().max())
    finally:
        N.lib().vs_debug_set(4, 0)
    # a window the LDS cannot hold (down-scaling by 12) takes the separate launches: same answer
    big = synthetic_frames(1, 256, 1300, seed=5).cuda()
    assert torch.equal(G.crop_resize_color(big, None, (20, 100), []), G.resize(big, (20, 100), True))
    with pytest.raises(N.NativeError):
        N.check(N.lib().vs_aug_crop_resize_color(N.ptr(x), N.ptr(x), 3, H, W, 0, 0, H + 1, W, 8, 8, 1, 0, None, None, N.stream()), "crop outside the frame")


def test_sequential_fuses_the_validation_chain_without_changing_values_or_random_draws():
    """augmentation/sequential.py:8-30 on the chain of BASELINE configs[2] (JPEG, Crop, Resize, Brightness, Contrast, Saturation, Hue): with the
    fused passes on and off the outputs are bit-identical -- image and mask, fixed strengths and random draws (same torch seed)"""
    x = synthetic_frames(4, 160, 144, seed=77).cuda()
    mask = (torch.rand(4, 1, 160, 144, generator=torch.Generator().manual_seed(3)) > 0.5).float().cuda()
    seq = G.Sequential(G.JPEG(), G.Crop(0.5, 0.9), G.Resize(0.6, 1.3), G.Brightness(0.5, 1.5), G.Contrast(0.5, 1.5), G.Saturation(0.5, 1.5), G.Hue(-0.1, 0.1))
    outs = {}
    for fuse in (True, False):
        G.FUSE = fuse
        try:
            res = []
            for args in ((40, 0.71, 0.71, 0.5, 1.5, 1.5, 0.1), (55,)):           # fixed strengths; only the JPEG quality given -> the rest drawn
                torch.manual_seed(11)
                res.append(seq(x, mask, args))
                torch.manual_seed(12)
                res.append(seq(x, None, args))
            outs[fuse] = res
        finally:
            G.FUSE = True
    torch.cuda.synchronize()
    for (ia, ma), (ib, mb) in zip(outs[True], outs[False]):
        assert ia.shape == ib.shape and torch.equal(ia, ib)
        assert (ma is None and mb is None) or torch.equal(ma, mb)
    assert outs[True][0][0].shape[-2:] == (int(0.71 * int(0.71 * 160)), int(0.71 * int(0.71 * 144)))
