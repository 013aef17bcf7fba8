#!/usr/bin/env python
"""Golden vectors for BASELINE configs[3] AT ITS STATED SIZE: the unmodified reference driven exactly as inference_streaming.py drives it.

    python tests/golden/make_golden_stream.py            # build container only (needs /root/reference); ~5-10 min of CPU

VideoSeal 1.0 (seed-0 state_dict of oracle/weights.py, loaded strictly into the reference module), a 128-frame 768 x 768 clip = 8 chunks
of 16 frames, for every chunk the two functions of the script, called as the script calls them:

  * `embed_video_clip(model, chunk_u8, msgs)` (inference_streaming.py:23-32): uint8 RGB24 -> /255 -> `model.embed(is_video=True,
    lowres_attenuation=True)` -> `(x * 255).byte()`;
  * `detect_video_clip(model, chunk_u8)` (inference_streaming.py:117-124) on the WATERMARKED uint8 chunk, the per-frame soft bits, and
    `soft_msgs.mean(0)` over the clip (inference_streaming.py:162-163);

and the same chunks as fp32 tensors (`model.embed` / `model.detect` without the byte round trip).  Stored: the logits in full, the
watermarked frames sub-sampled with float64 checksums, the aggregated decision.  tests/test_gpu_e2e.py compares the grouped streaming
path of this package (streaming.embed_detect_chunks with its defaults: key frames of 8 chunks per U-Net pass, 32 frames per extractor
pass, detect overlapped on a second stream -- what bench.py's stream leg times) against it.
"""
import importlib.util
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

from make_golden import REF, build_reference, import_reference, pack     # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs               # noqa: E402
from oracle.weights import make_state_dict, spec_from_card               # noqa: E402

N_FRAMES, SIZE, CHUNK, SEED = 128, 768, 16, 81


def streaming_functions():
    """embed_video_clip / detect_video_clip of the unmodified inference_streaming.py (its ffmpeg / tqdm imports are stubbed: only the two clip
    functions are used)"""
    import types
    for name in ("ffmpeg", "tqdm", "pytorch_msssim"):        # (evals/metrics.py:20 imports pytorch_msssim; the script uses bit_accuracy only)
        if name not in sys.modules and importlib.util.find_spec(name) is None:
            sys.modules[name] = types.ModuleType(name)
    import_reference()                      # the stub recipe (timm / torchvision / cv2 / av) + /root/reference on sys.path
    spec = importlib.util.spec_from_file_location("_ref_inference_streaming", os.path.join(REF, "inference_streaming.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.embed_video_clip, mod.detect_video_clip


@torch.no_grad()
def main():
    torch.manual_seed(0)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    embed_clip, detect_clip = streaming_functions()
    path = f"{REF}/videoseal/cards/videoseal_1.0.yaml"
    spec = spec_from_card(path)
    model = build_reference(spec, yaml.safe_load(open(path))).eval()
    print("strict load:", model.load_state_dict(make_state_dict(spec, seed=0), strict=True))
    frames = synthetic_frames(N_FRAMES, SIZE, SIZE, seed=SEED)
    msgs = synthetic_msgs(1, spec.nbits, seed=SEED)
    clip = (frames * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()          # RGB24, what ffmpeg hands the script
    t0 = time.time()
    w_u8, soft_u8, w_f32, preds_f32 = [], [], [], []
    for a in range(0, N_FRAMES, CHUNK):
        out = embed_clip(model, clip[a:a + CHUNK], msgs)
        w_u8.append(torch.from_numpy(out.copy()))
        soft_u8.append(detect_clip(model, out))
        wf = model.embed(frames[a:a + CHUNK], msgs=msgs, is_video=True, lowres_attenuation=True)["imgs_w"]
        w_f32.append(wf)
        preds_f32.append(model.detect(wf, is_video=True)["preds"])
        print(f"chunk {a // CHUNK}: {time.time() - t0:.0f} s", flush=True)
    w_u8, soft_u8, w_f32, preds_f32 = torch.cat(w_u8), torch.cat(soft_u8), torch.cat(w_f32), torch.cat(preds_f32)
    agg_u8, agg_f32 = soft_u8.mean(dim=0), preds_f32[:, 1:].mean(dim=0)
    delta = (255 * (w_f32 - frames)).double()
    psnr = 20 * np.log10(255.0) - 10 * np.log10(float((delta ** 2).mean()))
    d = {"meta": json.dumps(dict(name="vs10_stream_768", n=N_FRAMES, h=SIZE, w=SIZE, seed=SEED, chunk=CHUNK, model_chunk_size=int(model.chunk_size),
                                 step=int(model.step_size), psnr=psnr,
                                 bit_acc_u8=float(((agg_u8 > 0) == (msgs[0] > 0.5)).float().mean()),
                                 bit_acc_f32=float(((agg_f32 > 0) == (msgs[0] > 0.5)).float().mean())))}
    pack(d, "imgs_w", w_f32, stride=4099)
    d["w_u8.sub"] = w_u8.flatten()[::4099].numpy()
    d["w_u8.stats"] = np.array([float(w_u8.double().sum()), w_u8.numel(), 4099], dtype=np.float64)
    d["preds"] = preds_f32.numpy()
    d["soft_u8"] = soft_u8.numpy()
    d["agg_f32"], d["agg_u8"], d["msgs"] = agg_f32.numpy(), agg_u8.numpy(), msgs.numpy()
    np.savez_compressed(os.path.join(HERE, "vs10_stream_768.npz"), **d)
    print(d["meta"], f"min |agg| f32 {agg_f32.abs().min():.3e} u8 {agg_u8.abs().min():.3e}")


if __name__ == "__main__":
    main()
