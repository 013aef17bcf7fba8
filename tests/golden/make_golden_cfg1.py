#!/usr/bin/env python
"""Golden vectors at the STATED SIZE of BASELINE configs[1] and configs[4], produced by the UNMODIFIED reference.

    python tests/golden/make_golden_cfg1.py            # build container only (needs /root/reference); a few minutes of CPU
    python tests/golden/make_golden_cfg1.py --chunky   # configs[4] only (the 774 M-parameter extractor: ~8 GB of weights)

configs[1] -- VideoSeal 1.0 (seed-0 state_dict of oracle/weights.py, loaded strictly into the reference module), a batch of
32 frames of 768 x 768, image mode (`model.embed(imgs, msgs, is_video=False)`: one message per frame, every frame through the
U-Net, full-resolution JND, models/wam.py:134-204) then `model.detect` on the watermarked and on the clean frames
(models/wam.py:206-234) -- the workload `bench.py` times for its `value`.  Stored as `vs10_img_768x32.npz`: logits in full,
watermarked frames sub-sampled with float64 checksums, PSNR.

configs[4] -- the released ChunkySeal card (cards/chunkyseal.yaml:33-53: `convnext_chunky`, proportional_dim -> ConvNeXt dims
362 / 724 / 1448 / 2896, depths 3 / 3 / 27 / 3, stride-2 stem, 1024 bits) built by the reference's own `build_extractor`
(models/extractor.py:189-208), seed-2 state_dict, 2 frames of 1024 x 1024 through `Wam.detect`'s own resize to 256 x 256
(models/wam.py:206-234).  Stored as `chunky_detect_1024x2.npz`: logits `[2, 1025]` in full.  The embedder of the card is not
run at this size (configs[4] is the extractor), so only `detector.*` keys of the state_dict are loaded -- strictly.

tests/test_oracle_golden.py pins the CPU oracle on both fixtures, tests/test_gpu_e2e.py the HIP path.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, build_reference, import_reference, pack, toD     # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs               # noqa: E402
from oracle.weights import make_state_dict, spec_from_card               # noqa: E402

N_FRAMES, SIZE, SEED = 32, 768, 91
CH_FRAMES, CH_SIZE, CH_SEED, CH_SD_SEED = 2, 1024, 93, 2


@torch.no_grad()
def cfg1():
    path = f"{REF}/videoseal/cards/videoseal_1.0.yaml"
    spec = spec_from_card(path)
    model = build_reference(spec, yaml.safe_load(open(path))).eval()
    print("strict load:", model.load_state_dict(make_state_dict(spec, seed=0), strict=True))
    imgs = synthetic_frames(N_FRAMES, SIZE, SIZE, seed=SEED)
    msgs = synthetic_msgs(N_FRAMES, spec.nbits, seed=SEED)
    t0 = time.time()
    out = model.embed(imgs, msgs, is_video=False, lowres_attenuation=False)
    print(f"embed {time.time() - t0:.0f} s", flush=True)
    det = model.detect(out["imgs_w"], is_video=False)
    det_clean = model.detect(imgs, is_video=False)
    print(f"detect {time.time() - t0:.0f} s", flush=True)
    delta = (255 * (out["imgs_w"] - imgs)).double()
    psnr = 20 * np.log10(255.0) - 10 * np.log10(float((delta ** 2).mean()))
    psnr_frame = 20 * np.log10(255.0) - 10 * np.log10((delta ** 2).mean(dim=(1, 2, 3)).numpy())
    preds = det["preds"]
    bit_acc = float(((preds[:, 1:] > 0) == (msgs > 0.5)).float().mean())
    d = {"meta": json.dumps(dict(name="vs10_img_768x32", n=N_FRAMES, h=SIZE, w=SIZE, seed=SEED, is_video=False, lowres=False,
                                 chunk=int(model.chunk_size), step=int(model.step_size), video_mode="repeat", kind="smooth",
                                 psnr=psnr, bit_acc=bit_acc))}
    pack(d, "imgs_w", out["imgs_w"], stride=4099)
    pack(d, "preds_w", out["preds_w"], stride=131)
    d["preds"], d["preds_clean"], d["msgs"] = preds.numpy(), det_clean["preds"].numpy(), msgs.numpy()
    d["psnr_frame"] = psnr_frame
    np.savez_compressed(os.path.join(HERE, "vs10_img_768x32.npz"), **d)
    print(d["meta"], f"min |logit| {preds[:, 1:].abs().min():.3e} clean {det_clean['preds'][:, 1:].abs().min():.3e}")


@torch.no_grad()
def chunky():
    import_reference()
    from videoseal.models.extractor import build_extractor
    from videoseal.models.wam import Wam
    path = f"{REF}/videoseal/cards/chunkyseal.yaml"
    spec = spec_from_card(path)
    card = toD(yaml.safe_load(open(path)))
    ext = build_extractor(card.extractor.model, card.extractor.params, spec.img_size, spec.nbits)
    sd = make_state_dict(spec, seed=CH_SD_SEED)
    det_sd = {k[len("detector."):]: v for k, v in sd.items() if k.startswith("detector.")}
    del sd
    print("strict load (detector.*):", ext.load_state_dict(det_sd, strict=True), sum(v.numel() for v in det_sd.values()) / 1e6, "M parameters")
    del det_sd
    # Wam.detect (models/wam.py:206-234) needs only the detector, img_size and the interpolation defaults of the module
    model = Wam(torch.nn.Identity(), ext, torch.nn.Identity(), attenuation=None, scaling_w=spec.scaling_w, scaling_i=spec.scaling_i,
                img_size=spec.img_size).eval()
    imgs = synthetic_frames(CH_FRAMES, CH_SIZE, CH_SIZE, seed=CH_SEED)
    t0 = time.time()
    preds = model.detect(imgs)["preds"]
    print(f"chunky detect {time.time() - t0:.0f} s", tuple(preds.shape), flush=True)
    d = {"meta": json.dumps(dict(name="chunky_detect_1024x2", n=CH_FRAMES, h=CH_SIZE, w=CH_SIZE, seed=CH_SEED, sd_seed=CH_SD_SEED,
                                 nbits=spec.nbits, img_size=spec.img_size)),
         "preds": preds.numpy()}
    np.savez_compressed(os.path.join(HERE, "chunky_detect_1024x2.npz"), **d)
    print(d["meta"], f"|logit| median {preds[:, 1:].abs().median():.4f} min {preds[:, 1:].abs().min():.3e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    if "--chunky" not in sys.argv:
        cfg1()
    if "--cfg1" not in sys.argv:
        chunky()
