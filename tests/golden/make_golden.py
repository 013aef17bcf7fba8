#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

The reference modules are imported with in-memory stubs for the missing third
party packages; nothing from the reference is copied
into this repository.  The same seeded state_dict (oracle/weights.py) is loaded
strictly into the reference ``Videoseal`` module, so a mismatch in the key /
shape enumeration fails here.  Outputs are stored sub-sampled (plus float64
checksums) to keep the fixtures small.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle.inputs import synthetic_frames, synthetic_msgs          # noqa: E402
from oracle.weights import make_state_dict, spec_from_card, tiny_spec, legacy_tiny_spec, state_dict_layout   # noqa: E402


def import_reference():
    def stub(name, **kw):
        m = types.ModuleType(name); m.__dict__.update(kw); sys.modules[name] = m; return m

    class Inert:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return a[0] if a else None
        def __getattr__(self, k): return Inert()

    class DropPath(nn.Identity):
        pass

    stub("timm"); stub("timm.models"); stub("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=DropPath)
    tv = stub("torchvision")
    tvt = stub("torchvision.transforms", **{k: Inert for k in
               ("Compose", "ToTensor", "ToPILImage", "Resize", "CenterCrop", "ColorJitter", "RandomHorizontalFlip", "RandomCrop")})
    tv.transforms = tvt
    tvt.functional = stub("torchvision.transforms.functional")
    tv.utils = stub("torchvision.utils", save_image=Inert())
    stub("cv2"); stub("av")
    sys.path.insert(0, REF)
    from videoseal.models.embedder import build_embedder
    from videoseal.models.extractor import build_extractor
    from videoseal.models.videoseal import Videoseal
    from videoseal.augmentation.augmenter import get_dummy_augmenter
    from videoseal.modules.jnd import JND
    return build_embedder, build_extractor, Videoseal, get_dummy_augmenter, JND


class D(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def toD(x):
    return D({k: toD(v) for k, v in x.items()}) if isinstance(x, dict) else x


def build_reference(spec, card_like, device="cpu"):
    be, bx, Videoseal, dummy_aug, JND = import_reference()
    c = toD(card_like)
    emb = be(c.embedder.model, c.embedder.params, spec.nbits, spec.hidden / spec.nbits)
    ext = bx(c.extractor.model, c.extractor.params, spec.img_size, spec.nbits)
    return Videoseal(emb, ext, dummy_aug(), attenuation=(JND(in_channels=spec.jnd_in, out_channels=spec.jnd_out) if spec.jnd_in > 0 else None),
                     scaling_w=spec.scaling_w, scaling_i=spec.scaling_i, img_size=spec.img_size,
                     chunk_size=spec.chunk_size, step_size=spec.step_size)


def card_for_spec(s):
    """A card-shaped dict that makes the reference builders produce architecture ``s``."""
    pdec = {"pixelwise": False, "upscale_stages": [1], "embed_dim": s.dims[-1], "nbits": 16, "sigmoid_output": False}
    if s.extractor == "sam":
        ext = {"model": "sam_small",
               "params": {"encoder": {"img_size": s.img_size, "embed_dim": s.vit_dim, "out_chans": s.vit_out, "depth": s.vit_depth,
                                      "num_heads": s.vit_heads, "patch_size": s.vit_patch, "global_attn_indexes": list(s.vit_global),
                                      "window_size": s.vit_window, "mlp_ratio": s.vit_mlp_ratio, "qkv_bias": True,
                                      "use_rel_pos": s.vit_rel_pos},
                          "pixel_decoder": dict(pdec, embed_dim=s.vit_out, upscale_type="bilinear")}}
    else:
        ext = None
    return {
        "embedder": {"model": "unet_small2_yuv_quant" if s.yuv else "unet_rgb",
                     "params": {"msg_processor": {"nbits": 16, "hidden_size": 32, "msg_processor_type": "binary+concat"},
                                "unet": {"in_channels": s.in_ch, "out_channels": s.out_ch, "z_channels": s.z,
                                         "num_blocks": s.num_blocks, "activation": s.unet_act,
                                         "normalization": "rms" if s.unet_norm == "rms" else "batch",
                                         "z_channels_mults": list(s.mults), "last_tanh": s.last_tanh}}},
        "extractor": ext or {"model": "convnext_tiny",
                      "params": {"encoder": {"depths": list(s.depths), "dims": list(s.dims), "stem_stride": s.stem_stride},
                                 "pixel_decoder": {"pixelwise": False, "upscale_stages": [1], "embed_dim": s.dims[-1],
                                                   "nbits": 16, "sigmoid_output": False}}},
    }


def sub(t, stride=7):
    """strided sub-sample + float64 checksums of a tensor."""
    flat = t.detach().double().flatten()
    return {"sub": t.detach().flatten()[::stride].float().numpy(), "sum": float(flat.sum()),
            "sumsq": float((flat ** 2).sum()), "numel": int(flat.numel())}


def pack(d, prefix, t, stride=7):
    s = sub(t, stride)
    d[prefix + ".sub"] = s["sub"]
    d[prefix + ".stats"] = np.array([s["sum"], s["sumsq"], s["numel"], stride], dtype=np.float64)


@torch.no_grad()
def run_case(model, spec, name, *, n, h, w, seed, is_video, lowres, chunk=None, step=None, video_mode="repeat",
             kind="smooth", detect_interp=None):
    imgs = synthetic_frames(n, h, w, seed=seed, kind=kind)
    msgs = synthetic_msgs(1 if is_video else n, spec.nbits, seed=seed)
    if chunk: model.chunk_size = chunk
    if step: model.step_size = step
    model.video_mode = video_mode
    out = model.embed(imgs, msgs, is_video=is_video, lowres_attenuation=lowres)
    det = model.detect(out["imgs_w"], is_video=is_video)
    det_clean = model.detect(imgs, is_video=is_video)
    msg_hat = model.extract_message(out["imgs_w"]) if is_video else None
    delta = (255 * (out["imgs_w"] - imgs)).double()
    psnr = 20 * np.log10(255.0) - 10 * np.log10(float((delta ** 2).mean()))
    d = {"meta": json.dumps(dict(name=name, n=n, h=h, w=w, seed=seed, is_video=is_video, lowres=lowres,
                                 chunk=model.chunk_size, step=model.step_size, video_mode=video_mode, kind=kind,
                                 psnr=psnr))}
    pack(d, "imgs_w", out["imgs_w"])
    if "preds_w" in out:
        pack(d, "preds_w", out["preds_w"])
    d["preds"] = det["preds"].numpy()
    d["preds_clean"] = det_clean["preds"].numpy()
    d["msgs"] = msgs.numpy()
    if msg_hat is not None:
        d["msg_hat"] = msg_hat.numpy()
    # sub-module outputs (model.embedder / model.detector contract)
    if h == spec.img_size and w == spec.img_size and not is_video:
        x = imgs
        y = model.rgb2yuv(x)[:, 0:1] if model.embedder.yuv else x
        pack(d, "delta", model.embedder(y, msgs))
        if model.attenuation is not None:
            pack(d, "hmaps", model.attenuation.heatmaps(x))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: psnr={psnr:.3f} dB  bit_acc={( (det['preds'][:,1:]>0) == (msgs>0.5)).float().mean():.3f}"
          f"  |logit| median={det['preds'][:,1:].abs().median():.4f} min={det['preds'][:,1:].abs().min():.2e}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    # ---- key/shape lists of the real reference for every released card (meta device: no allocation)
    keys = {}
    for card in ("videoseal_1.0", "pixelseal", "chunkyseal", "videoseal_0.0"):
        path = f"{REF}/videoseal/cards/{card}.yaml"
        spec = spec_from_card(path)
        _ls = torch.linspace                      # convnext.py:122 calls .item() on a linspace: keep that one on cpu
        torch.linspace = lambda *a, **k: _ls(*a, **{**k, "device": "cpu"})
        try:
            with torch.device("meta"):
                m = build_reference(spec, yaml.safe_load(open(path)))
        finally:
            torch.linspace = _ls
        ref_keys = {k: list(v.shape) for k, v in m.state_dict().items()}
        mine = {k: list(v) for k, v in state_dict_layout(spec).items()}
        assert ref_keys == mine, (card, set(ref_keys) ^ set(mine))
        assert list(ref_keys) == list(mine), f"{card}: key ORDER differs"
        keys[card] = ref_keys
        print(card, len(ref_keys), "tensors,", sum(int(np.prod(v)) for v in ref_keys.values()) / 1e6, "M elements")
    json.dump(keys, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=0)

    if "--legacy-only" in sys.argv:
        return legacy_cases()
    # ---- VideoSeal 1.0 (full size), seed 0
    path = f"{REF}/videoseal/cards/videoseal_1.0.yaml"
    spec = spec_from_card(path)
    model = build_reference(spec, yaml.safe_load(open(path))).eval()
    sd = make_state_dict(spec, seed=0)
    missing = model.load_state_dict(sd, strict=True)
    print("strict load:", missing)
    run_case(model, spec, "vs10_img256", n=1, h=256, w=256, seed=1, is_video=False, lowres=False)
    run_case(model, spec, "vs10_img_odd", n=2, h=200, w=328, seed=2, is_video=False, lowres=False)
    run_case(model, spec, "vs10_img_lowres", n=1, h=300, w=280, seed=3, is_video=False, lowres=True)
    run_case(model, spec, "vs10_vid", n=10, h=144, w=176, seed=4, is_video=True, lowres=False, chunk=2, step=4)
    run_case(model, spec, "vs10_vid_lowres", n=9, h=288, w=352, seed=5, is_video=True, lowres=True, chunk=32, step=4)
    run_case(model, spec, "vs10_img_uniform", n=1, h=256, w=256, seed=6, is_video=False, lowres=False, kind="uniform")

    # ---- tiny architecture (fast CPU tests), seed 3
    ts = tiny_spec()
    tm = build_reference(ts, card_for_spec(ts)).eval()
    tm.load_state_dict(make_state_dict(ts, seed=3), strict=True)
    run_case(tm, ts, "tiny_img", n=3, h=64, w=64, seed=11, is_video=False, lowres=False)
    run_case(tm, ts, "tiny_img_resize", n=2, h=90, w=130, seed=12, is_video=False, lowres=False)
    run_case(tm, ts, "tiny_vid_repeat", n=11, h=80, w=72, seed=13, is_video=True, lowres=False, chunk=2, step=2)
    run_case(tm, ts, "tiny_vid_alternate", n=9, h=64, w=64, seed=14, is_video=True, lowres=True, chunk=4, step=3,
             video_mode="alternate")
    run_case(tm, ts, "tiny_vid_interpolate", n=9, h=70, w=66, seed=15, is_video=True, lowres=False, chunk=3, step=2,
             video_mode="interpolate")

    # ---- ChunkySeal-shaped tiny architecture: RGB in/out embedder, stem stride 2, channel counts that are not multiples
    # of 4 (18, 54, 90) and odd feature maps (31 -> 15 -> 7 -> 3), seed 4
    tc = tiny_spec(yuv=False, in_ch=3, out_ch=3, dims=[18, 36, 54, 90], stem_stride=2, hidden=32, nbits=16)
    tcm = build_reference(tc, card_for_spec(tc)).eval()
    tcm.load_state_dict(make_state_dict(tc, seed=4), strict=True)
    run_case(tcm, tc, "tinyc_img", n=2, h=64, w=64, seed=21, is_video=False, lowres=False)
    run_case(tcm, tc, "tinyc_vid", n=7, h=96, w=80, seed=22, is_video=True, lowres=True, chunk=2, step=2)


def legacy_cases():
    """videoseal_0.0 card: RMSNorm/SiLU RGB U-Net, SAM-style ViT extractor with windowed + global attention and
    decomposed relative positions, no JND, scaling_w = 1."""
    path = f"{REF}/videoseal/cards/videoseal_0.0.yaml"
    spec = spec_from_card(path)
    model = build_reference(spec, yaml.safe_load(open(path))).eval()
    print("strict load:", model.load_state_dict(make_state_dict(spec, seed=5), strict=True))
    run_case(model, spec, "vs00_img256", n=2, h=256, w=256, seed=31, is_video=False, lowres=False)
    run_case(model, spec, "vs00_vid", n=6, h=144, w=176, seed=32, is_video=True, lowres=False, chunk=2, step=2)
    tv = legacy_tiny_spec()
    tm = build_reference(tv, card_for_spec(tv)).eval()
    tm.load_state_dict(make_state_dict(tv, seed=6), strict=True)
    run_case(tm, tv, "tinyv_img", n=3, h=64, w=64, seed=41, is_video=False, lowres=False)
    run_case(tm, tv, "tinyv_img_resize", n=2, h=90, w=130, seed=42, is_video=False, lowres=False)
    run_case(tm, tv, "tinyv_vid", n=7, h=80, w=72, seed=43, is_video=True, lowres=False, chunk=2, step=2)


if __name__ == "__main__":
    main()
    if "--legacy-only" not in sys.argv:
        legacy_cases()
