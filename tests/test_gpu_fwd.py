"""The training forward (Wam.forward / Videoseal.video_forward) on the HIP path against fixtures produced by the reference module
in train mode (tests/golden/make_golden_fwd.py): un-attenuated preds_w, attenuation(imgs, imgs_w), resized imgs_aug, the
Augmenter's seeded picks, batch-statistics BatchNorm with its running-statistics update; plus the PixelSeal card, the median
filter against utils/image.py's output, a load() round trip and BASELINE configs[2] at its stated size."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment as A  # noqa: E402
from oracle import videoseal_ref as R  # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs  # noqa: E402
from oracle.weights import make_state_dict, spec_from_card, tiny_spec  # noqa: E402
from tests._util import GOLDEN, assert_decisions, check_sub, load_golden  # noqa: E402

FWD_MARGIN = 2e-5         # train-mode forward: measured logit error 1.3e-6 (gpurun_out/r06a/decisions.jsonl -> profiles/r06a_decision_margins.txt)
CHAIN_MARGIN = 2e-5       # through the augmentation chain: measured logit error 2.2e-6 (profiles/r06a_decision_margins.txt)
from tests.test_gpu_e2e import _run_case, make_model  # noqa: E402
from tests.test_oracle_fwd import FWD_FULL, FWD_TINY, PIXELSEAL, bn_vectors  # noqa: E402
from tests.test_oracle_golden import CARDS  # noqa: E402

import videoseal_amd  # noqa: E402
from videoseal_amd import augmentation as G  # noqa: E402


def hip_forward(model, spec, meta):
    imgs = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"], kind=meta["kind"])
    msgs = synthetic_msgs(1 if meta["is_video"] else meta["n"], spec.nbits, seed=meta["seed"])
    masks = torch.ones(meta["n"], 1, meta["h"], meta["w"])
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs=meta["augs"], augs_params=meta["augs_params"], num_augs=meta["num_augs"])
    model.train()
    if not meta["bn_train"]:
        model.embedder.eval()
        model.detector.eval()
    model.step_size, model.video_mode, model.lowres_attenuation = meta["step"], meta["video_mode"], meta["lowres"]
    model.blender.scaling_i = meta["scaling_i"]
    torch.manual_seed(meta["torch_seed"])
    return model(imgs.cuda(), masks.cuda(), msgs, is_video=meta["is_video"]), imgs, msgs


def _check_fwd(spec, sd, name):
    g = load_golden(name)
    meta = g["meta"]
    model = make_model(spec, sd)           # fresh module: train-mode BatchNorm updates its running statistics
    out, imgs, msgs = hip_forward(model, spec, meta)
    assert out["selected_aug"] == meta["selected_aug"], "the Augmenter did not pick like the reference"
    assert list(out["imgs_aug"].shape) == meta["aug_shape"] and list(out["masks"].shape) == meta["mask_shape"]
    assert abs(float(out["masks"].mean()) - meta["mask_mean"]) < 1e-6
    assert set(out) == ({"msgs", "masks", "imgs_w", "imgs_aug", "preds", "selected_aug"} | (set() if meta["is_video"] else {"preds_w"}))
    check_sub(g, "imgs_w", out["imgs_w"], 1e-4, name + " ")
    check_sub(g, "imgs_aug", out["imgs_aug"], 1e-4, name + " ")
    if "preds_w.sub" in g:
        check_sub(g, "preds_w", out["preds_w"], 1e-4, name + " ")
    gold = torch.from_numpy(g["preds"])
    preds = out["preds"].cpu()
    assert (preds - gold).abs().max() < 1e-3
    assert_decisions(preds, gold, margin=FWD_MARGIN, what=name + " train-mode forward", min_sure=0.99)
    assert (out["msgs"].cpu().numpy() == g["msgs"]).all()
    # BatchNorm buffers after the call: updated in train mode (momentum 0.1, unbiased variance), untouched otherwise
    rm, rv, nbt = bn_vectors({k: v.cpu() for k, v in model.state_dict().items()}, None)
    grm, grv = torch.from_numpy(g["bn_running_mean"]), torch.from_numpy(g["bn_running_var"])
    assert (rm - grm).abs().max() < 2e-5
    assert ((rv - grv).abs() / grv.abs().clamp_min(1e-3)).max() < 1e-4
    assert (nbt.numpy() == g["bn_nbt"]).all()
    if meta["bn_train"]:
        # the folded (eval) weights must be re-packed from the UPDATED running statistics
        model.eval()
        model.blender.scaling_i = spec.scaling_i
        new_sd = {k: v.cpu() for k, v in model.state_dict().items()}
        w = model.embed(imgs.cuda(), msgs if not meta["is_video"] else msgs, is_video=meta["is_video"])["imgs_w"].cpu()
        if meta["is_video"]:
            ref = R.embed_video(new_sd, spec, imgs, msgs, step_size=meta["step"], chunk_size=model.chunk_size, video_mode=meta["video_mode"])
        else:
            ref = R.embed_image(new_sd, spec, imgs, msgs)
        assert (w - ref["imgs_w"]).abs().max() < 1e-4
        old = R.embed_image(sd, spec, imgs[:1], msgs[:1] if not meta["is_video"] else msgs)["imgs_w"]
        assert (ref["imgs_w"][:1] - old).abs().max() > 1e-5 or meta["is_video"]      # the statistics did move


@pytest.fixture(scope="module")
def tiny_sd():
    s = tiny_spec()
    return s, make_state_dict(s, seed=3)


@pytest.mark.parametrize("name", FWD_TINY)
def test_tiny_forward_matches_reference(tiny_sd, name):
    _check_fwd(*tiny_sd, name)


@pytest.mark.parametrize("name", FWD_FULL)
def test_vs10_train_forward_matches_reference(name):
    s = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
    _check_fwd(s, make_state_dict(s, seed=0), name)


def test_forward_batch_of_videos_and_asserts(tiny_sd):
    spec, sd = tiny_sd
    model = make_model(spec, sd)
    model.step_size = 2
    x = synthetic_frames(8, 64, 64, seed=5).view(2, 4, 3, 64, 64).cuda()
    outs = model(x, torch.ones(2, 4, 1, 64, 64).cuda(), None, is_video=True)          # videoseal.py:142-153: list of per-video dicts
    assert isinstance(outs, list) and len(outs) == 2 and outs[0]["imgs_w"].shape == (4, 3, 64, 64) and outs[0]["msgs"].shape == (4, spec.nbits)
    with pytest.raises(AssertionError):
        model(x[0, 0], None, None, is_video=True)
    with pytest.raises(AssertionError):
        model(x, None, None, is_video=False)
    with pytest.raises(AssertionError, match="unique"):
        model.video_forward(x[0], None, synthetic_msgs(2, spec.nbits))


def test_mask_blend_and_mask_embedders(tiny_sd):
    """augmenter.py:171-176: imgs_w * m + imgs * (1 - m) with caller-supplied masks (the segmentation branch of the mixed embedder)"""
    spec, sd = tiny_sd
    model = make_model(spec, sd)
    model.augmenter = G.Augmenter(masks={"kind": "given"}, augs={"identity": 1}, augs_params={})
    model.train(); model.embedder.eval(); model.detector.eval()
    imgs = synthetic_frames(2, 64, 64, seed=9)
    masks = (torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(1)) > 0.4).float()
    msgs = synthetic_msgs(2, spec.nbits, seed=9)
    out = model(imgs.cuda(), masks.cuda(), msgs, is_video=False)
    ref = R.forward_image(sd, spec, imgs, masks, msgs, lambda iw, im, mk, v, r: (iw * mk + im * (1 - mk), mk, "Identity"))
    assert (out["imgs_aug"].cpu() - ref["imgs_aug"]).abs().max() < 1e-4 and torch.equal(out["masks"].cpu(), masks)
    assert (out["preds"].cpu() - ref["preds"]).abs().max() < 1e-3
    assert torch.equal(out["imgs_aug"].cpu()[masks.expand(-1, 3, -1, -1) == 0], imgs[masks.expand(-1, 3, -1, -1) == 0])
    with pytest.raises(NotImplementedError):       # the OpenCV 'mixed' embedder (kind None) is loud, not silently full
        G.Augmenter(masks={"kind": None}, augs={"identity": 1}, augs_params={})(imgs.cuda(), imgs.cuda(), None)


@pytest.mark.parametrize("name", PIXELSEAL)
def test_pixelseal_matches_reference_golden(name):
    """released PixelSeal card: the wider Y-channel U-Net (z 32..256, bottleneck 512 + 256 message channels), step 8"""
    s = spec_from_card(os.path.join(CARDS, "pixelseal.yaml"))
    sd = make_state_dict(s, seed=7)
    _run_case(s, sd, make_model(s, sd), name)


def test_augmenter_picks_match_reference():
    """names + output shapes (which encode the crop-size / offset draws) of 12 seeded sequences of the REFERENCE Augmenter"""
    rows = json.load(open(os.path.join(GOLDEN, "augmenter_picks.json")))
    aug = G.Augmenter(masks={"kind": "none"}, augs={"identity": 2, "crop": 3, "hflip": 1},
                      augs_params={"crop": {"min_size": 0.5, "max_size": 0.9}}, num_augs=3).train()
    for r in rows:
        torch.manual_seed(r["seed"])
        x = torch.zeros(2, 3, 60, 84, device="cuda")
        y, m, names = aug(x, x, None, is_video=bool(r["seed"] & 1), do_resize=bool(r["seed"] & 2))
        assert names == r["names"] and list(y.shape) == r["shape"] and list(m.shape) == r["mask_shape"], r


def test_median_filter_matches_reference_utils_image():
    z = np.load(os.path.join(GOLDEN, "median_ref.npz"))
    x = torch.from_numpy(z["x"]).cuda()
    for k in (3, 5, 7):
        assert torch.equal(G.median_filter(x, k).cpu(), torch.from_numpy(z[f"k{k}"])), k


def test_gaussian_noise_drop_frame_speed_change():
    x = synthetic_frames(6, 40, 48, seed=2).cuda()
    torch.manual_seed(4)
    y, _ = G.GaussianNoise(0.01, 0.1)(x, None)
    torch.manual_seed(4)
    std = torch.rand(1).item() * 0.09 + 0.01          # valuemetric.py:182-185, then randn_like on the image's device
    assert torch.equal(y, x + torch.randn_like(x) * std)
    import random
    random.seed(3)
    y, _ = G.DropFrame(0.5)(x, None)
    random.seed(3)
    ref = x.clone()
    for i in range(6):                                 # video.py:518-526
        if random.random() >= 0.5:
            continue
        ref[i] = x[(i + (-1 if random.random() < 0.5 else 1)) % 6]
    assert torch.equal(y, ref)
    y, m = G.SpeedChange()(x, torch.ones(6, 1, 40, 48).cuda(), 0.5)
    idx = torch.linspace(0, 5, 12).round().long()
    assert torch.equal(y, x[idx]) and m.shape[0] == 12
    assert torch.equal(G.SpeedChange()(x, None, 2.0)[0], x[torch.linspace(0, 5, 12)[:6].round().long()])
    assert "drop_frame" in G.name2aug and "gaussian_noise" in G.name2aug


def test_load_round_trip(tmp_path, monkeypatch):
    """videoseal.load(card) with a checkpoint file on disk: torch.save({'model': sd}) -> load -> strict=False load -> same outputs"""
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    card = {"checkpoint_path": str(tmp_path / "ckpt.pth"),
            "args": {"attenuation": "jnd_1_1", "nbits": spec.nbits, "hidden_size_multiplier": spec.hidden / spec.nbits, "img_size_proc": spec.img_size,
                     "blending_method": "additive", "scaling_w": spec.scaling_w, "scaling_i": spec.scaling_i,
                     "videowam_chunk_size": spec.chunk_size, "videowam_step_size": spec.step_size},
            "embedder": {"model": "unet_small2_yuv_quant",
                         "params": {"msg_processor": {"nbits": 16, "hidden_size": 32, "msg_processor_type": "binary+concat"},
                                    "unet": {"in_channels": spec.in_ch, "out_channels": spec.out_ch, "z_channels": spec.z,
                                             "num_blocks": spec.num_blocks, "activation": "relu", "normalization": "batch",
                                             "z_channels_mults": list(spec.mults), "last_tanh": spec.last_tanh}}},
            "extractor": {"model": "convnext_tiny",
                          "params": {"encoder": {"depths": list(spec.depths), "dims": list(spec.dims), "stem_stride": spec.stem_stride},
                                     "pixel_decoder": {"pixelwise": False, "upscale_stages": [1], "embed_dim": spec.dims[-1],
                                                       "nbits": 16, "sigmoid_output": False}}}}
    import yaml
    (tmp_path / "tinycard.yaml").write_text(yaml.safe_dump(card))
    extra = dict(sd)
    extra["some.unknown.key"] = torch.zeros(3)                # strict=False: unknown keys are reported, not fatal (cfg.py:147-150)
    torch.save({"model": extra}, tmp_path / "ckpt.pth")
    model = videoseal_amd.load(tmp_path / "tinycard.yaml")
    assert model.training and model.device.type == "cpu"      # like the reference: CPU, train mode
    model = model.eval().to("cuda")
    imgs = synthetic_frames(2, 64, 64, seed=8)
    msgs = synthetic_msgs(2, spec.nbits, seed=8)
    w = model.embed(imgs.cuda(), msgs, is_video=False)["imgs_w"].cpu()
    assert (w - R.embed_image(sd, spec, imgs, msgs)["imgs_w"]).abs().max() < 1e-4
    with pytest.raises(FileNotFoundError):
        videoseal_amd.load("no_such_card")


def test_config3_full_size_vs10_clip_through_the_chain():
    """BASELINE configs[2] as written: VideoSeal 1.0, 16-frame 768x768 clip -> embed -> Sequential(JPEG, Crop, Resize, Brightness,
    Contrast, Saturation, Hue) at the fixed strengths of augmentation/__init__.py:107-123 -> detect, vs the oracle chain (Pillow JPEG,
    ATen resize, restated colour ops) applied to OUR watermarked frames."""
    spec = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
    sd = make_state_dict(spec, seed=0)
    model = make_model(spec, sd)
    imgs = synthetic_frames(16, 768, 768, seed=303)
    msgs = synthetic_msgs(1, spec.nbits, seed=303)
    w = model.embed(imgs.cuda(), msgs, is_video=True)["imgs_w"]
    chain = G.Sequential(G.JPEG(), G.Crop(), G.Resize(), G.Brightness(), G.Contrast(), G.Saturation(), G.Hue())
    args = (40, 0.71, 0.71, 0.5, 1.5, 1.5, 0.1)
    torch.manual_seed(21)
    aug, _ = chain(w, None, args)
    wc = w.cpu()
    torch.manual_seed(21)
    r = A.jpeg(wc, 40)
    th = tw = int(0.71 * 768)
    i = torch.randint(0, 768 - th + 1, size=(1,)).item(); j = torch.randint(0, 768 - tw + 1, size=(1,)).item()
    r = A.crop(r, i, j, th, tw)
    r = A.resize(r, (int(0.71 * th), int(0.71 * tw)))
    r = A.hue(A.saturation(A.contrast(A.brightness(r, 0.5), 1.5), 1.5), 0.1)
    assert aug.shape == r.shape and (aug.cpu() - r).abs().max() < 1e-5
    preds = model.detect(aug, is_video=True)["preds"].cpu()
    with torch.no_grad():
        pref = R.detect(sd, spec, r)["preds"]
    assert (preds - pref).abs().max() < 1e-3
    assert_decisions(preds, pref, margin=CHAIN_MARGIN, what="configs[2] chain vs oracle", min_sure=0.99)
    m = msgs.expand(16, -1).float()
    assert (R.bit_accuracy(preds[:, 1:], m) - R.bit_accuracy(pref[:, 1:], m)).abs().max() < 1e-3
    # aggregated decision over the clip (videoseal.py:411-428)
    agg, agg_ref = preds[:, 1:].mean(0), pref[:, 1:].mean(0)
    assert ((agg > 0) == (agg_ref > 0))[agg_ref.abs() > 2e-5].all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_model_on_a_non_default_device(tiny_sd):
    """kernels must launch on the stream of the MODEL's device, whatever the current device is"""
    spec, sd = tiny_sd
    m = make_model(spec, sd).to("cuda:1")
    imgs = synthetic_frames(2, 64, 64, seed=8)
    msgs = synthetic_msgs(2, spec.nbits, seed=8)
    with torch.cuda.device(0):
        w = m.embed(imgs.to("cuda:1"), msgs, is_video=False)["imgs_w"]
        p = m.detect(w, is_video=False)["preds"]
    assert w.device.index == 1 and (w.cpu() - R.embed_image(sd, spec, imgs, msgs)["imgs_w"]).abs().max() < 1e-4 and p.shape[0] == 2
