"""Pin the CPU oracle (oracle/videoseal_ref.py) against golden vectors produced by the
UNMODIFIED reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import pytest
import torch

from oracle import videoseal_ref as R
from oracle.inputs import synthetic_frames, synthetic_msgs
from oracle.weights import legacy_tiny_spec, make_state_dict, spec_from_card, state_dict_layout, tiny_spec
from tests._util import GOLDEN, check_sub, load_golden, psnr_np

CARDS = os.path.join(os.path.dirname(GOLDEN), "..", "videoseal_amd", "cards")

TINY = ["tiny_img", "tiny_img_resize", "tiny_vid_repeat", "tiny_vid_alternate", "tiny_vid_interpolate"]
TINYC = ["tinyc_img", "tinyc_vid"]
FULL = ["vs10_img256", "vs10_img_odd", "vs10_img_lowres", "vs10_vid", "vs10_vid_lowres", "vs10_img_uniform"]


def _run(spec, sd, meta):
    imgs = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"], kind=meta["kind"])
    msgs = synthetic_msgs(1 if meta["is_video"] else meta["n"], spec.nbits, seed=meta["seed"])
    if meta["is_video"]:
        out = R.embed_video(sd, spec, imgs, msgs, lowres_attenuation=meta["lowres"], chunk_size=meta["chunk"],
                            step_size=meta["step"], video_mode=meta["video_mode"])
    else:
        out = R.embed_image(sd, spec, imgs, msgs, lowres_attenuation=meta["lowres"])
    return imgs, msgs, out


@pytest.fixture(scope="module")
def tiny():
    s = tiny_spec()
    return s, make_state_dict(s, seed=3)


@pytest.fixture(scope="module")
def vs10():
    s = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
    return s, make_state_dict(s, seed=0)


def _check_case(spec, sd, name):
    g = load_golden(name)
    meta = g["meta"]
    with torch.no_grad():
        imgs, msgs, out = _run(spec, sd, meta)
        assert (msgs.numpy() == g["msgs"]).all()
        check_sub(g, "imgs_w", out["imgs_w"], 2e-6, name + " ")
        if "preds_w.sub" in g:
            check_sub(g, "preds_w", out["preds_w"], 2e-6, name + " ")
        preds = R.detect(sd, spec, out["imgs_w"])["preds"]
        assert (preds - torch.from_numpy(g["preds"])).abs().max() < 2e-5
        assert ((preds > 0).numpy() == (g["preds"] > 0)).all(), "bit decisions differ from the reference"
        clean = R.detect(sd, spec, imgs)["preds"]
        assert (clean - torch.from_numpy(g["preds_clean"])).abs().max() < 2e-5
        assert abs(psnr_np(out["imgs_w"], imgs) - meta["psnr"]) < 1e-3
        if meta["is_video"]:
            mh = R.extract_message(sd, spec, out["imgs_w"])
            assert (mh.numpy() == g["msg_hat"]).all()
        if "delta.sub" in g:
            y = R.rgb2y(sd, imgs) if spec.yuv else imgs
            check_sub(g, "delta", R.embedder_forward(sd, spec, y, msgs), 2e-6, name + " ")
            if "hmaps.sub" in g:
                check_sub(g, "hmaps", R.jnd_heatmaps(sd, spec, imgs), 1e-6, name + " ")


@pytest.mark.parametrize("name", TINY)
def test_tiny_oracle_matches_reference(tiny, name):
    _check_case(*tiny, name)


@pytest.fixture(scope="module")
def tinyc():
    s = tiny_spec(yuv=False, in_ch=3, out_ch=3, dims=[18, 36, 54, 90], stem_stride=2, hidden=32, nbits=16)
    return s, make_state_dict(s, seed=4)


@pytest.mark.parametrize("name", TINYC)
def test_tiny_chunky_oracle_matches_reference(tinyc, name):
    _check_case(*tinyc, name)


@pytest.mark.parametrize("name", FULL)
def test_vs10_oracle_matches_reference(vs10, name):
    _check_case(*vs10, name)


@pytest.fixture(scope="module")
def tinyv():
    s = legacy_tiny_spec()
    return s, make_state_dict(s, seed=6)


@pytest.mark.parametrize("name", ["tinyv_img", "tinyv_img_resize", "tinyv_vid"])
def test_tiny_legacy_oracle_matches_reference(tinyv, name):
    """videoseal_0.0 family: RMSNorm/SiLU U-Net + ViT extractor (windowed / global attention, relative positions)"""
    _check_case(*tinyv, name)


@pytest.mark.parametrize("name", ["vs00_img256", "vs00_vid"])
def test_vs00_oracle_matches_reference(name):
    s = spec_from_card(os.path.join(CARDS, "videoseal_0.0.yaml"))
    _check_case(s, make_state_dict(s, seed=5), name)


def test_state_dict_layout_matches_reference_cards():
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    for card, ref in keys.items():
        spec = spec_from_card(os.path.join(CARDS, card + ".yaml"))
        mine = {k: list(v) for k, v in state_dict_layout(spec).items()}
        assert mine == ref and list(mine) == list(ref), card
    assert len(keys["videoseal_1.0"]) == 434


def test_metrics():
    a = torch.rand(2, 3, 8, 8)
    b = (a + 0.01).clamp(0, 1)
    p = R.psnr(a, b)
    assert p.shape == (2,) and (p > 30).all()
    assert R.psnr(a, b, is_video=True).ndim == 0
    preds = torch.tensor([[0.3, -0.2, 1.0, -5.0]])
    assert R.bit_accuracy(preds, torch.tensor([[1, 0, 0, 0]])).item() == 0.75


def test_vs10_oracle_matches_the_reference_streaming_run_at_768(vs10):
    """configs[3] at its stated frame size: the oracle's per-chunk `embed_video(lowres_attenuation=True)` + `detect` on the first two 16-frame
    chunks of the 128-frame 768 x 768 fixture (tests/golden/make_golden_stream.py: the unmodified reference through inference_streaming.py's
    own clip functions) -- logits of those 32 frames and the strided sample of their watermarked pixels"""
    spec, sd = vs10
    g = load_golden("vs10_stream_768")
    meta = g["meta"]
    frames = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"])[:32]
    msgs = synthetic_msgs(1, spec.nbits, seed=meta["seed"])
    stride = int(g["imgs_w.stats"][3])
    per_frame = 3 * meta["h"] * meta["w"]
    for c in range(2):
        w = R.embed_video(sd, spec, frames[16 * c:16 * c + 16], msgs, lowres_attenuation=True, chunk_size=meta["model_chunk_size"], step_size=meta["step"])["imgs_w"]
        first = 16 * c * per_frame
        k0 = -(-first // stride)                                     # first sample index that falls into this chunk
        k1 = -(-(first + 16 * per_frame) // stride)
        mine = w.flatten()[k0 * stride - first::stride]
        assert mine.numel() == k1 - k0
        assert (mine - torch.from_numpy(g["imgs_w.sub"][k0:k1])).abs().max().item() < 2e-6
        preds = R.detect(sd, spec, w)["preds"]
        assert (preds - torch.from_numpy(g["preds"][16 * c:16 * c + 16])).abs().max().item() < 2e-5


def test_vs10_oracle_matches_the_reference_at_configs1_stated_size(vs10):
    """configs[1] at its stated size (tests/golden/make_golden_cfg1.py: the unmodified reference on 32 frames of 768 x 768, image mode,
    full-resolution JND): frames of an image-mode batch are independent (eval BatchNorm), so the oracle runs the first 6 of the 32 --
    the strided sample of their watermarked pixels, their logits on watermarked and clean frames, decisions identical"""
    spec, sd = vs10
    g = load_golden("vs10_img_768x32")
    meta = g["meta"]
    assert (meta["n"], meta["h"], meta["w"], meta["is_video"], meta["lowres"]) == (32, 768, 768, False, False)
    nf = 6
    imgs = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"])[:nf]
    msgs = synthetic_msgs(meta["n"], spec.nbits, seed=meta["seed"])
    assert (msgs.numpy() == g["msgs"]).all()
    with torch.no_grad():
        w = R.embed_image(sd, spec, imgs, msgs[:nf], lowres_attenuation=False)["imgs_w"]
        stride = int(g["imgs_w.stats"][3])
        k1 = -(-(nf * 3 * meta["h"] * meta["w"]) // stride)
        assert (w.flatten()[::stride] - torch.from_numpy(g["imgs_w.sub"][:k1])).abs().max().item() < 2e-6
        d = (255.0 * (w.double() - imgs.double()))
        psnr_frame = 20 * torch.log10(torch.tensor(255.0, dtype=torch.float64)) - 10 * torch.log10((d ** 2).mean(dim=(1, 2, 3)))
        assert (psnr_frame - torch.from_numpy(g["psnr_frame"][:nf])).abs().max().item() < 1e-3
        for frames, key in ((w, "preds"), (imgs, "preds_clean")):
            p = R.detect(sd, spec, frames)["preds"]
            gold = torch.from_numpy(g[key][:nf])
            assert (p - gold).abs().max().item() < 2e-5
            assert ((p > 0) == (gold > 0)).all(), "bit decisions differ from the reference"


def test_chunkyseal_oracle_matches_the_reference_at_its_released_size():
    """configs[4]: the released ChunkySeal extractor (ConvNeXt 362 / 724 / 1448 / 2896, depths 3 / 3 / 27 / 3, stride-2 stem, 1024 bits;
    773.7 M parameters) built by the reference's own `build_extractor`, 2 frames of 1024 x 1024 through `Wam.detect`
    (tests/golden/make_golden_cfg1.py --chunky) -- until round 6 the oracle was pinned for this architecture only on the 18 / 36 / 54 /
    90-channel `tinyc` goldens"""
    g = load_golden("chunky_detect_1024x2")
    meta = g["meta"]
    spec = spec_from_card(os.path.join(CARDS, "chunkyseal.yaml"))
    assert (spec.nbits, spec.img_size) == (meta["nbits"], meta["img_size"]) == (1024, 256)
    sd = {k: v for k, v in make_state_dict(spec, seed=meta["sd_seed"]).items() if k.startswith("detector.")}
    imgs = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"])
    with torch.no_grad():
        p = R.detect(sd, spec, imgs)["preds"]
    gold = torch.from_numpy(g["preds"])
    assert p.shape == gold.shape == (2, 1025)
    assert (p - gold).abs().max().item() < 2e-5
    assert ((p > 0) == (gold > 0)).all(), "bit decisions differ from the reference"
