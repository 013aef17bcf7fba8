"""Pin the oracle's TRAINING forward (batch-statistics BatchNorm, forward()/video_forward() semantics), its Augmenter pick
logic, the median filter and the PixelSeal architecture against fixtures produced by the unmodified reference
(tests/golden/make_golden_fwd.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import augment as A
from oracle import videoseal_ref as R
from oracle.inputs import synthetic_frames, synthetic_msgs
from oracle.weights import make_state_dict, spec_from_card, tiny_spec
from tests._util import GOLDEN, check_sub, load_golden
from tests.test_oracle_golden import CARDS, _check_case

FWD_TINY = ["tiny_fwd_img_train", "tiny_fwd_img_evalbn", "tiny_fwd_img_si", "tiny_fwd_vid_train", "tiny_fwd_vid_lowres", "tiny_fwd_vid_alt"]
FWD_FULL = ["vs10_fwd_img_train"]
PIXELSEAL = ["pixelseal_img", "pixelseal_vid"]


def oracle_forward(spec, sd, meta):
    imgs = synthetic_frames(meta["n"], meta["h"], meta["w"], seed=meta["seed"], kind=meta["kind"])
    msgs = synthetic_msgs(1 if meta["is_video"] else meta["n"], spec.nbits, seed=meta["seed"])
    masks = torch.ones(meta["n"], 1, meta["h"], meta["w"])
    aug = A.Augmenter(meta["augs"], meta["augs_params"], meta["num_augs"])
    bn = {} if meta["bn_train"] else None
    torch.manual_seed(meta["torch_seed"])
    with torch.no_grad():
        if meta["is_video"]:
            out = R.forward_video(sd, spec, imgs, masks, msgs, aug, bn=bn, step_size=meta["step"], video_mode=meta["video_mode"],
                                  lowres_attenuation=meta["lowres"])
        else:
            out = R.forward_image(sd, spec, imgs, masks, msgs, aug, bn=bn, scaling_i=meta["scaling_i"])
    return out, bn


def bn_vectors(sd, bn):
    merged = dict(sd)
    merged.update(bn or {})
    rm = torch.cat([v.flatten() for k, v in merged.items() if k.endswith("running_mean")])
    rv = torch.cat([v.flatten() for k, v in merged.items() if k.endswith("running_var")])
    nbt = torch.stack([v for k, v in merged.items() if k.endswith("num_batches_tracked")])
    return rm, rv, nbt


def _check_fwd(spec, sd, name):
    g = load_golden(name)
    meta = g["meta"]
    out, bn = oracle_forward(spec, sd, meta)
    assert out["selected_aug"] == meta["selected_aug"]
    assert list(out["imgs_aug"].shape) == meta["aug_shape"] and list(out["masks"].shape) == meta["mask_shape"]
    assert abs(float(out["masks"].mean()) - meta["mask_mean"]) < 1e-6
    check_sub(g, "imgs_w", out["imgs_w"], 2e-6, name + " ")
    check_sub(g, "imgs_aug", out["imgs_aug"], 2e-6, name + " ")
    if "preds_w.sub" in g:
        check_sub(g, "preds_w", out["preds_w"], 5e-6, name + " ")
    assert (out["preds"] - torch.from_numpy(g["preds"])).abs().max() < 5e-5
    assert (out["msgs"].numpy() == g["msgs"]).all()
    rm, rv, nbt = bn_vectors(sd, bn)
    assert (rm - torch.from_numpy(g["bn_running_mean"])).abs().max() < 1e-6
    assert ((rv - torch.from_numpy(g["bn_running_var"])).abs() / torch.from_numpy(g["bn_running_var"]).abs().clamp_min(1e-3)).max() < 1e-5
    assert (nbt.numpy() == g["bn_nbt"]).all()


@pytest.fixture(scope="module")
def tiny():
    s = tiny_spec()
    return s, make_state_dict(s, seed=3)


@pytest.mark.parametrize("name", FWD_TINY)
def test_tiny_forward_matches_reference(tiny, name):
    _check_fwd(*tiny, name)


@pytest.mark.parametrize("name", FWD_FULL)
def test_vs10_train_forward_matches_reference(name):
    s = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
    _check_fwd(s, make_state_dict(s, seed=0), name)


@pytest.mark.parametrize("name", PIXELSEAL)
def test_pixelseal_oracle_matches_reference(name):
    s = spec_from_card(os.path.join(CARDS, "pixelseal.yaml"))
    _check_case(s, make_state_dict(s, seed=7), name)


def test_augmenter_picks_match_reference():
    """names AND output shapes (which encode the crop-size draws) of 12 seeded 3-op sequences of the reference Augmenter"""
    rows = json.load(open(os.path.join(GOLDEN, "augmenter_picks.json")))
    aug = A.Augmenter({"identity": 2, "crop": 3, "hflip": 1}, {"crop": {"min_size": 0.5, "max_size": 0.9}}, 3)
    for r in rows:
        torch.manual_seed(r["seed"])
        x = torch.zeros(2, 3, 60, 84)
        y, m, names = aug(x, x, None, is_video=bool(r["seed"] & 1), do_resize=bool(r["seed"] & 2))
        assert names == r["names"] and list(y.shape) == r["shape"] and list(m.shape) == r["mask_shape"], r


def test_median_filter_matches_reference_utils_image():
    """oracle.augment.median_filter vs videoseal/utils/image.py:60-84 run by make_golden_fwd.py (ties included)"""
    z = np.load(os.path.join(GOLDEN, "median_ref.npz"))
    x = torch.from_numpy(z["x"])
    for k in (3, 5, 7):
        assert torch.equal(A.median_filter(x, k), torch.from_numpy(z[f"k{k}"])), k
