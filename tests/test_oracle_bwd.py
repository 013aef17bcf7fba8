"""Pin the oracle of the training BACKWARD (oracle/loss.py + autograd through oracle/videoseal_ref.py's functional
forward) against fixtures produced by the unmodified reference's train-mode forward, its own `VideosealLoss` and `loss.backward()`
(tests/golden/make_golden_bwd.py).  CPU only.  These fixtures are the target the HIP backward kernels are compared with."""
import numpy as np
import pytest
import torch

from oracle import augment as A
from oracle import loss as L
from oracle import videoseal_ref as R
from oracle.inputs import synthetic_frames, synthetic_msgs
import os

from oracle.weights import make_state_dict, spec_from_card, tiny_spec
from tests._util import load_golden, projection_vector
from tests.test_oracle_golden import CARDS

CASES = ["tiny_bwd_img_recipe", "tiny_bwd_img_balanced", "tiny_bwd_vid_recipe", "vs10_bwd_img_recipe"]


def oracle_step(spec, sd0, meta, names):
    """one accumulation step of train.py:626-643 on the oracle: returns (log, {name: grad})"""
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in names:
        sd[k].requires_grad_(True)
    n, h, w = meta["n"], meta["h"], meta["w"]
    imgs = synthetic_frames(n, h, w, seed=meta["seed"], kind=meta["kind"])
    msgs = synthetic_msgs(1 if meta["is_video"] else n, spec.nbits, seed=meta["seed"])
    masks = torch.ones(n, 1, h, w)
    aug = A.Augmenter(meta["augs"], meta["augs_params"], meta["num_augs"])
    torch.manual_seed(meta["torch_seed"])
    if meta["is_video"]:
        out = R.forward_video(sd, spec, imgs, masks, msgs, aug, bn={}, step_size=meta["step"])
    else:
        out = R.forward_image(sd, spec, imgs, masks, msgs, aug, bn={})
    assert out["selected_aug"] == meta["selected_aug"]
    preds = out["preds"] / meta["temperature"]
    total, log = L.videoseal_loss(imgs, out["imgs_w"], out["masks"], out["msgs"], preds, last_layer=sd[meta["last_layer"]],
                                  **{k: v for k, v in meta["loss_kw"].items()})
    (total / meta["accumulation"]).backward()
    return preds.detach(), log, {k: sd[k].grad for k in names}


@pytest.mark.parametrize("name", CASES)
def test_oracle_backward_matches_reference(name):
    if name.startswith("vs10"):          # the released architecture at its working size (339 trainable tensors)
        spec = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml"))
        sd = make_state_dict(spec, seed=0)
    else:
        spec = tiny_spec()
        sd = make_state_dict(spec, seed=3)
    g = load_golden(name)
    meta = g["meta"]
    names = [str(k) for k in g["grad_names"]]
    assert meta["no_grad_params"] == []
    preds, log, grads = oracle_step(spec, sd, meta, names)
    assert (preds - torch.from_numpy(g["preds"])).abs().max() < 5e-5
    for k, v in meta["log"].items():
        assert abs(float(log[k]) - v) <= 2e-5 * max(1.0, abs(v)), (k, float(log[k]), v)
    ref = g["grad_summary"]
    gmax = ref[:, 0].max()
    for i, k in enumerate(names):
        gr = grads[k]
        assert gr is not None, k
        gd = gr.double().flatten()
        got = np.array([float(gd.norm()), float(gd.sum()), float((gd * projection_vector(k, gd.numel())).sum())])
        # absolute tolerance scaled to the tensor's own gradient norm (sum / projection of numel terms), floor for vanishing gradients
        tol = 5e-5 * max(ref[i, 0], 1e-4 * gmax) * max(1.0, np.sqrt(gd.numel()) / 16)
        assert np.all(np.abs(got - ref[i]) <= tol), (k, got, ref[i], tol)
    for k in [str(k) for k in g["full_names"]]:
        rf = torch.from_numpy(g["grad." + k])
        assert (grads[k] - rf).abs().max() <= 5e-5 * rf.abs().max() + 1e-9, k


def test_adaptive_scales_follow_the_reference_formula():
    """videosealloss.py:72-107 on a hand-made graph: two losses reaching `last_layer` with known gradient norms"""
    w = torch.tensor([3.0, 4.0], requires_grad=True)
    l1 = (w * torch.tensor([1.0, 0.0])).sum() * 2.0          # grad (2, 0), norm 2
    l2 = (w * torch.tensor([3.0, 4.0])).sum()                # grad (3, 4), norm 5
    s = L.adaptive_scales([l1, l2], [1.0, 3.0], w)            # N = norm of the last = 5
    assert abs(float(s[0]) - 0.25 * 5 / 2) < 1e-6 and abs(float(s[1]) - 0.75) < 1e-6
    s = L.adaptive_scales([l1, l2], [1.0, 3.0], w, total_norm=10.0)
    assert abs(float(s[0]) - 0.25 * 10 / 2) < 1e-6 and abs(float(s[1]) - 0.75 * 10 / 5) < 1e-6
    l3 = torch.tensor(1.0)                                    # does not reach the layer: zero gradient (videosealloss.py:86-88)
    s = L.adaptive_scales([l3, l2], [1.0, 1.0], w)
    assert float(s[0]) > 1e11 and abs(float(s[1]) - 0.5) < 1e-6


def test_decoding_loss_per_pixel_branch_uses_masked_pixels_only():
    """videosealloss.py:157-169"""
    torch.manual_seed(0)
    preds = torch.randn(2, 1 + 5, 4, 6)
    msgs = torch.randint(0, 2, (2, 5))
    masks = torch.zeros(2, 1, 4, 6)
    masks[:, :, 1:3, 2:5] = 1
    got = L.decoding_loss(preds, msgs, masks)
    sel = preds[:, 1:, 1:3, 2:5]
    ref = torch.nn.functional.binary_cross_entropy_with_logits(sel, msgs[:, :, None, None].expand_as(sel).float())
    assert abs(float(got) - float(ref)) < 1e-6
