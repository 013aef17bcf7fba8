#!/usr/bin/env python
"""Golden vectors for the TRAINING forward (Wam.forward / Videoseal.video_forward, wam.py:68-132, videoseal.py:163-256), the
Augmenter's seeded picks (augmenter.py:137-152), the PixelSeal card and utils/image.py's median filter, produced by running
the UNMODIFIED reference modules in this container (needs /root/reference):

    python tests/golden/make_golden_fwd.py

Same import recipe as make_golden.py (in-memory stubs for the packages that are not installed).  Three torchvision
functions are given working stand-ins because the Augmenter cases call them; each is pure indexing or a published RNG
order, restated here (torchvision is neither installed nor vendored by the reference):
  transforms.functional.crop(img, i, j, h, w) -> img[..., i:i+h, j:j+w]        (in-bounds crops only)
  transforms.functional.hflip(img)            -> img.flip(-1)
  transforms.RandomCrop.get_params(img, (th, tw)): (0, 0, h, w) if the sizes match, else i = randint(0, h-th+1), j =
                                                  randint(0, w-tw+1) drawn in that order with torch.randint(size=(1,))
Everything else (the model, the Augmenter, the mask embedder, BatchNorm in train mode) is the reference's own code.
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                   # noqa: E402

from oracle.inputs import synthetic_frames, synthetic_msgs          # noqa: E402
from oracle.weights import make_state_dict, spec_from_card, tiny_spec   # noqa: E402


class _RandomCrop:
    def __init__(self, *a, **k):
        pass

    @staticmethod
    def get_params(img, output_size):
        h, w = img.shape[-2:]
        th, tw = output_size
        if h < th or w < tw:
            raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
        if w == tw and h == th:
            return 0, 0, h, w
        i = torch.randint(0, h - th + 1, size=(1,)).item()
        j = torch.randint(0, w - tw + 1, size=(1,)).item()
        return i, j, th, tw


def patch_torchvision():
    import torchvision.transforms as T
    import torchvision.transforms.functional as TF
    TF.crop = lambda img, i, j, h, w: img[..., i:i + h, j:j + w]
    TF.hflip = lambda img: img.flip(-1)
    T.RandomCrop = _RandomCrop


def bn_state(model):
    sd = model.state_dict()
    rm = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("running_mean")])
    rv = torch.cat([v.flatten() for k, v in sd.items() if k.endswith("running_var")])
    nbt = torch.stack([v for k, v in sd.items() if k.endswith("num_batches_tracked")])
    return rm.clone(), rv.clone(), nbt.clone()


AUGS = {"identity": 1, "crop": 3, "hflip": 1}
AUG_PARAMS = {"crop": {"min_size": 0.5, "max_size": 0.9}}


def run_forward(model, Augmenter, spec, name, *, n, h, w, seed, is_video, bn_train, lowres=False, step=None,
                video_mode="repeat", num_augs=2, scaling_i=None, kind="smooth"):
    imgs = synthetic_frames(n, h, w, seed=seed, kind=kind)
    msgs = synthetic_msgs(1 if is_video else n, spec.nbits, seed=seed)
    masks = torch.ones(n, 1, h, w)
    model.augmenter = Augmenter(masks={"kind": "none"}, augs=dict(AUGS), augs_params=dict(AUG_PARAMS), num_augs=num_augs)
    model.train()
    if not bn_train:                 # BatchNorm on its running statistics, augmenter still in its training branch
        model.embedder.eval()
        model.detector.eval()
    if step:
        model.step_size = step
    model.video_mode = video_mode
    model.lowres_attenuation = lowres
    if scaling_i is not None:
        model.blender.scaling_i = scaling_i
    torch.manual_seed(1000 + seed)
    with torch.no_grad():
        out = model(imgs, masks, msgs, is_video=is_video)
    d = {"meta": json.dumps(dict(name=name, n=n, h=h, w=w, seed=seed, is_video=is_video, bn_train=bn_train, lowres=lowres,
                                 step=model.step_size, video_mode=video_mode, num_augs=num_augs, kind=kind,
                                 scaling_i=float(model.blender.scaling_i), selected_aug=out["selected_aug"],
                                 aug_shape=list(out["imgs_aug"].shape), mask_shape=list(out["masks"].shape),
                                 mask_mean=float(out["masks"].mean()), augs=AUGS, augs_params=AUG_PARAMS, torch_seed=1000 + seed))}
    MG.pack(d, "imgs_w", out["imgs_w"], 5)
    MG.pack(d, "imgs_aug", out["imgs_aug"], 5)
    if "preds_w" in out:
        MG.pack(d, "preds_w", out["preds_w"], 5)
    d["preds"] = out["preds"].numpy()
    d["msgs"] = out["msgs"].numpy()
    rm, rv, nbt = bn_state(model)
    d["bn_running_mean"], d["bn_running_var"], d["bn_nbt"] = rm.numpy(), rv.numpy(), nbt.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: aug={out['selected_aug']} aug_shape={tuple(out['imgs_aug'].shape)} |preds| max={out['preds'].abs().max():.3f} "
          f"nbt={int(nbt[0])}")
    model.blender.scaling_i = spec.scaling_i


def augmenter_picks(Augmenter):
    """seeded pick sequences of the reference Augmenter alone (names + output shapes): augmenter.py:137-152 + the ops' own draws"""
    aug = Augmenter(masks={"kind": "none"}, augs={"identity": 2, "crop": 3, "hflip": 1}, augs_params=dict(AUG_PARAMS), num_augs=3)
    aug.train()
    rows = []
    for seed in range(12):
        torch.manual_seed(seed)
        x = torch.zeros(2, 3, 60, 84)
        y, m, names = aug(x, x, None, is_video=bool(seed & 1), do_resize=bool(seed & 2))
        rows.append({"seed": seed, "names": names, "shape": list(y.shape), "mask_shape": list(m.shape)})
    return rows


def main():
    torch.set_num_threads(8)
    MG.import_reference()
    patch_torchvision()
    from videoseal.augmentation.augmenter import Augmenter
    from videoseal.utils.image import median_filter

    # ---- tiny architecture, seed 3 (fresh module per case: train-mode BN updates the running statistics)
    ts = tiny_spec()

    def tiny_model():
        m = MG.build_reference(ts, MG.card_for_spec(ts))
        m.load_state_dict(make_state_dict(ts, seed=3), strict=True)
        return m
    run_forward(tiny_model(), Augmenter, ts, "tiny_fwd_img_train", n=4, h=72, w=88, seed=31, is_video=False, bn_train=True)
    run_forward(tiny_model(), Augmenter, ts, "tiny_fwd_img_evalbn", n=3, h=64, w=64, seed=32, is_video=False, bn_train=False)
    run_forward(tiny_model(), Augmenter, ts, "tiny_fwd_img_si", n=2, h=80, w=64, seed=33, is_video=False, bn_train=False, scaling_i=0.9)
    run_forward(tiny_model(), Augmenter, ts, "tiny_fwd_vid_train", n=6, h=80, w=72, seed=34, is_video=True, bn_train=True, step=2)
    run_forward(tiny_model(), Augmenter, ts, "tiny_fwd_vid_lowres", n=7, h=90, w=70, seed=35, is_video=True, bn_train=False, step=3,
                lowres=True, video_mode="interpolate")
    run_forward(tiny_model(), Augmenter, ts, "tiny_fwd_vid_alt", n=5, h=64, w=64, seed=36, is_video=True, bn_train=True, step=2,
                video_mode="alternate", num_augs=1)

    # ---- VideoSeal 1.0 at full size, train-mode BN
    path = f"{MG.REF}/videoseal/cards/videoseal_1.0.yaml"
    spec = spec_from_card(path)
    model = MG.build_reference(spec, yaml.safe_load(open(path)))
    model.load_state_dict(make_state_dict(spec, seed=0), strict=True)
    run_forward(model, Augmenter, spec, "vs10_fwd_img_train", n=2, h=300, w=280, seed=41, is_video=False, bn_train=True)

    # ---- PixelSeal (wider U-Net, step 8): embed / detect in eval mode like make_golden.run_case
    path = f"{MG.REF}/videoseal/cards/pixelseal.yaml"
    ps = spec_from_card(path)
    pm = MG.build_reference(ps, yaml.safe_load(open(path))).eval()
    pm.load_state_dict(make_state_dict(ps, seed=7), strict=True)
    MG.run_case(pm, ps, "pixelseal_img", n=1, h=256, w=256, seed=51, is_video=False, lowres=False)
    MG.run_case(pm, ps, "pixelseal_vid", n=9, h=200, w=240, seed=52, is_video=True, lowres=True, chunk=32, step=8)

    # ---- Augmenter picks + median filter of utils/image.py:60-84
    json.dump(augmenter_picks(Augmenter), open(os.path.join(HERE, "augmenter_picks.json"), "w"), indent=0)
    g = torch.Generator().manual_seed(99)
    x = torch.rand(2, 3, 40, 52, generator=g)
    x[:, :, 10:20, 5:30] = (x[:, :, 10:20, 5:30] * 4).round() / 4          # ties
    np.savez_compressed(os.path.join(HERE, "median_ref.npz"), x=x.numpy(), **{f"k{k}": median_filter(x, k).numpy() for k in (3, 5, 7)})
    print("augmenter picks + median fixtures written")


if __name__ == "__main__":
    main()
