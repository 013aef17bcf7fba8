#!/usr/bin/env python
"""Calibrates bench.py's `cpu_baseline` (kind "port": the oracle restatement, the only CPU path that can travel to the GPU box) against the
UNMODIFIED reference module on the same host: both run VideoSeal 1.0 embed + detect on the same 4 frames of 768x768 (image mode, eval),
same threads.  Needs /root/reference (this container).  Writes profiles/r03_cpu_port_vs_reference.json, which bench.py quotes.

    python tools/cpu_port_vs_reference.py"""
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG                                   # noqa: E402

from oracle import videoseal_ref as R                      # noqa: E402
from oracle.inputs import synthetic_frames, synthetic_msgs  # noqa: E402
from oracle.weights import make_state_dict, spec_from_card  # noqa: E402


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.time(); fn(); ts.append(time.time() - t0)
    return min(ts)


def main():
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    MG.import_reference()
    path = f"{MG.REF}/videoseal/cards/videoseal_1.0.yaml"
    spec = spec_from_card(path)
    sd = make_state_dict(spec, seed=0)
    model = MG.build_reference(spec, yaml.safe_load(open(path)))
    model.load_state_dict(sd, strict=True)
    model.eval()
    n, size = 4, 768
    imgs = synthetic_frames(n, size, size, seed=0, kind="uniform")
    msgs = synthetic_msgs(n, spec.nbits)

    def ref():
        with torch.no_grad():
            w = model.embed(imgs, msgs, is_video=False)["imgs_w"]
            model.detect(w, is_video=False)

    def port():
        with torch.no_grad():
            w = R.embed_image(sd, spec, imgs, msgs)["imgs_w"]
            R.detect(sd, spec, w)
    tr, tp = best(ref), best(port)
    out = {"reference_frames_per_s": round(n / tr, 3), "port_frames_per_s": round(n / tp, 3), "port_over_reference": round(tr / tp, 3),
           "threads": threads, "host_cpus": os.cpu_count(),
           "note": f"measured in the build container ({os.cpu_count()} CPUs, {threads} threads): unmodified reference module {n / tr:.2f} frames/s, oracle port "
                   f"{n / tp:.2f} frames/s on {n} frames {size}x{size}, image mode, embed + detect (tools/cpu_port_vs_reference.py)"}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03_cpu_port_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
