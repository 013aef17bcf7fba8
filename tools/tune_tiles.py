#!/usr/bin/env python
"""Write videoseal_amd/tuned_tiles_gfx950.json: per-conv-signature tile choices for the benchmark configurations (BASELINE configs
2 image / 2 video / 4 stream at 32 resp. 16 frames per call, ChunkySeal detect), each candidate timed over many interleaved rounds.
GPU box only:  python tools/tune_tiles.py  (then copy gpurun_out/tuned_tiles_gfx950.json into videoseal_amd/)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VIDEOSEAL_TUNED_TILES"] = "0"
import torch
import videoseal_amd

# start from the packaged table: the signatures of the other arithmetic (VIDEOSEAL_CONV=bf16x3 / f16x2: the key ends in 2 for
# 2 x f16) are kept, the ones of the arithmetic in use are re-measured
from videoseal_amd.engine import TUNED_TILES
cache, kept = {}, {}
if os.path.exists(TUNED_TILES) and "--fresh" not in sys.argv:
    h2 = os.environ.get("VIDEOSEAL_CONV", "split") != "bf16x3"
    for k, v in json.load(open(TUNED_TILES)).items():
        kt = tuple(json.loads(k))
        if (len(kt) == 22) != h2:
            kept[kt] = v
def run(card, B, S, video, lowres, detect_only=False, rounds=8):
    model = videoseal_amd.build(card, seed=0).eval().cuda()
    model.chunk_size = max(model.chunk_size, B)
    eng = model._engine()
    eng.tune_rounds = rounds
    eng._tile_cache.update(cache)
    x = torch.rand(B, 3, S, S, device="cuda")
    msgs = torch.randint(0, 2, (1 if video else B, model.embedder.cfg.nbits))
    if detect_only:
        model.detect(x, is_video=True)
    else:
        w = model.embed(x, msgs, is_video=video, lowres_attenuation=lowres)["imgs_w"]
        model.detect(w, is_video=True)
    torch.cuda.synchronize()
    cache.update(eng._tile_cache)
    print(card, B, S, video, "->", len(cache), "signatures", flush=True)
    del model
    torch.cuda.empty_cache()

run("videoseal_1.0", 32, 768, False, False)
run("videoseal_1.0", 32, 768, True, False)
run("videoseal_1.0", 16, 768, True, True)
run("videoseal_1.0", 16, 768, True, False)      # chain mode (BASELINE configs[2]): 16-frame clip, full-resolution JND
run("videoseal_1.0", 1, 256, False, False)
if "--chunky" in sys.argv:
    run("chunkyseal", 16, 1024, False, False, detect_only=True, rounds=3)
out = os.path.join(ROOT, "gpurun_out", "tuned_tiles_gfx950.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    cache.update(kept)
    json.dump({json.dumps(list(k)): v for k, v in sorted(cache.items(), key=lambda kv: (len(kv[0]), kv[0]))}, f, indent=0)
print("wrote", out)
