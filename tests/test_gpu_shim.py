"""The reference's own caller code against the `videoseal` import shim: the clip functions of inference_streaming.py:23-32 and
119-125 (numpy uint8 RGB24 chunk in, CPU tensors through model.embed / model.detect, numpy out) and the README quick start."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import videoseal  # noqa: E402  (the shim package at the repo root)
from videoseal.evals.metrics import bit_accuracy  # noqa: E402
from videoseal.models import Videoseal  # noqa: E402

from oracle import videoseal_ref as R  # noqa: E402
from oracle.inputs import synthetic_frames  # noqa: E402
from oracle.weights import make_state_dict, tiny_spec  # noqa: E402
from tests.test_gpu_e2e import make_model  # noqa: E402


def embed_video_clip(model: Videoseal, clip: np.ndarray, msgs: torch.Tensor) -> np.ndarray:
    """the call sequence of inference_streaming.py:23-32"""
    clip_tensor = torch.tensor(clip, dtype=torch.float32).permute(0, 3, 1, 2) / 255.0
    outputs = model.embed(clip_tensor, msgs=msgs, is_video=True, lowres_attenuation=True)
    processed_clip = outputs["imgs_w"]
    return (processed_clip * 255.0).byte().permute(0, 2, 3, 1).numpy()


def detect_video_clip(model: Videoseal, clip: np.ndarray) -> torch.Tensor:
    """the call sequence of inference_streaming.py:119-125"""
    clip_tensor = torch.tensor(clip, dtype=torch.float32).permute(0, 3, 1, 2) / 255.0
    outputs = model.detect(clip_tensor, is_video=True)
    return outputs["preds"][:, 1:]


def test_inference_streaming_clip_functions_run_unchanged():
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    model = make_model(spec, sd)
    assert isinstance(model, Videoseal) and isinstance(model, videoseal.models.Wam)
    model.chunk_size, model.step_size = 8, 2
    clip = (synthetic_frames(16, 96, 128, seed=61) * 255).byte().permute(0, 2, 3, 1).numpy()
    torch.manual_seed(0)
    msgs = model.get_random_msg()
    out = embed_video_clip(model, clip, msgs)
    assert isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.shape == clip.shape
    # the same through the CPU oracle (reference arithmetic): uint8 frames equal except where x*255 sits within rounding noise of an integer
    x = torch.tensor(clip, dtype=torch.float32).permute(0, 3, 1, 2) / 255.0
    ref = R.embed_video(sd, spec, x, msgs, lowres_attenuation=True, chunk_size=8, step_size=2)["imgs_w"]
    ref_u8 = (ref * 255.0).byte().permute(0, 2, 3, 1).numpy()
    diff = np.abs(out.astype(np.int16) - ref_u8.astype(np.int16))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    # and the fused uint8 entry point produces exactly what the caller's fp32 round trip produces
    u8 = model.embed_u8(torch.from_numpy(clip).cuda(), msgs)["imgs_w"].cpu().numpy()
    assert np.array_equal(u8, out)
    bits = detect_video_clip(model, out)
    assert bits.shape == (16, spec.nbits) and bits.device.type == "cpu"
    pref = R.detect(sd, spec, torch.tensor(out, dtype=torch.float32).permute(0, 3, 1, 2) / 255.0)["preds"][:, 1:]
    assert (bits - pref).abs().max() < 1e-3
    acc = bit_accuracy(bits, msgs.expand(16, -1)).mean()
    assert abs(float(acc) - float(R.bit_accuracy(pref, msgs.expand(16, -1)).mean())) < 1e-3
    # inference_streaming.py:160: mean of the logits over the clip, > 0
    assert ((bits.mean(0) > 0) == (pref.mean(0) > 0))[pref.mean(0).abs() > 2e-5].all()


def test_readme_quick_start_calls():
    """README.md:61-72: load -> embed(imgs, is_video=False) -> detect -> (preds[:, 1:] > 0)"""
    spec = tiny_spec()
    model = make_model(spec, make_state_dict(spec, seed=3))
    img = synthetic_frames(1, 120, 100, seed=62).cuda()
    outputs = model.embed(img, is_video=False)
    imgs_w, msgs = outputs["imgs_w"], outputs["msgs"]
    assert msgs.shape == (1, spec.nbits) and msgs.dtype == torch.int64
    detected = model.detect(imgs_w, is_video=False)
    hidden = (detected["preds"][0, 1:] > 0).float()
    assert hidden.shape == (spec.nbits,)
    from videoseal.evals.metrics import psnr
    assert float(psnr(imgs_w, img)) > 25


def test_model_built_by_the_train_py_factories_runs_like_the_card_built_one():
    """train.py:262-305: build_embedder + build_extractor + Videoseal(...) -> the same HIP engine configuration as the card path: with one
    state_dict both models return bit-identical frames and logits; a train-mode forward of the factory-built model carries gradients to the
    optimizer's parameter list (train.py:330, 626-643)"""
    from videoseal.augmentation.augmenter import get_dummy_augmenter
    from videoseal.models import build_embedder, build_extractor
    from videoseal.modules.jnd import JND
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    ref_model = make_model(spec, sd)
    emb_cfg = {"msg_processor": {"nbits": 16, "hidden_size": 32, "msg_processor_type": "binary+concat"},
               "unet": {"in_channels": spec.in_ch, "out_channels": spec.out_ch, "z_channels": spec.z, "num_blocks": spec.num_blocks, "activation": "relu",
                        "normalization": "batch", "z_channels_mults": list(spec.mults), "last_tanh": spec.last_tanh}}
    ext_cfg = {"encoder": {"depths": list(spec.depths), "dims": list(spec.dims)}, "pixel_decoder": {"upscale_stages": [1], "nbits": 16}}
    embedder = build_embedder("unet_tiny_yuv_quant" if spec.yuv else "unet_tiny_quant", emb_cfg, spec.nbits, spec.hidden / spec.nbits)
    extractor = build_extractor("convnext_tiny", ext_cfg, spec.img_size, spec.nbits)
    wam = Videoseal(embedder, extractor, get_dummy_augmenter(), JND(in_channels=1, out_channels=1), spec.scaling_w, spec.scaling_i,
                    img_size=spec.img_size, chunk_size=spec.chunk_size, step_size=spec.step_size, blending_method="additive", lowres_attenuation=False)
    msg = wam.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    wam = wam.to("cuda").eval()
    x = synthetic_frames(6, 112, 96, seed=63).cuda()
    msgs = torch.randint(0, 2, (1, spec.nbits), generator=torch.Generator().manual_seed(1))
    a = ref_model.embed(x, msgs, is_video=True)["imgs_w"]
    b = wam.embed(x, msgs, is_video=True)["imgs_w"]
    assert torch.equal(a, b)
    assert torch.equal(ref_model.detect(a, is_video=True)["preds"], wam.detect(b, is_video=True)["preds"])
    wam.train()
    params = list(embedder.parameters()) + list(extractor.parameters())
    out = wam(x[:2], torch.ones(2, 1, 112, 96, device="cuda"), torch.randint(0, 2, (2, spec.nbits)).cuda(), is_video=False)
    loss = out["preds"][:, 1:].square().mean() + (out["imgs_w"] - x[:2]).square().mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params if p.requires_grad)
