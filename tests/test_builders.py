"""train.py's model-construction path (train.py:262-305) against this package: `build_embedder` / `build_extractor` with the
reference's signatures and side effects, configs keyed by model name, `Videoseal(embedder, extractor, augmenter, attenuation, ...)`,
the optimizer's parameter list, and the `videoseal` import shim as an overlay over a reference checkout.  CPU only."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests._util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

# (card whose key list the construction must reproduce, embedder model, extractor model, nbits, hidden_size_multiplier, img_size_proc, attenuation)
CASES = [("videoseal_1.0", "unet_small2_yuv_quant", "convnext_tiny", 256, 1, 256, "jnd_1_1"),
         ("pixelseal", "unet_base_yuv_quant", "convnext_tiny", 256, 1, 256, "jnd_1_1"),
         ("chunkyseal", "unet_chunky", "convnext_chunky", 1024, 2, 256, "jnd_1_1"),
         ("videoseal_0.0", None, "sam_small", 96, 2, 256, "none")]


def _card_args(card):
    import yaml
    with open(os.path.join(ROOT, "videoseal_amd", "cards", card + ".yaml")) as f:
        return yaml.safe_load(f)


@pytest.mark.parametrize("card,emb_name,ext_name,nbits,mult,img_size,att", CASES)
def test_train_py_construction_path(card, emb_name, ext_name, nbits, mult, img_size, att):
    """the statements of train.py:262-305 + 330, with `videoseal.models` resolved by the shim and the configs of this package"""
    from videoseal.augmentation.augmenter import Augmenter
    from videoseal.models import Videoseal, build_embedder, build_extractor
    from videoseal.modules.jnd import JND
    from videoseal_amd.builders import load_config, to_attrdict
    c = _card_args(card)
    a = c["args"]
    assert int(a["nbits"]) == nbits and a.get("hidden_size_multiplier", 2) == mult
    if emb_name is None:                 # the legacy card's embedder is not in configs/embedder.yaml of the reference either: take the card's sub-tree
        emb_name, embedder_params = c["embedder"]["model"], to_attrdict(c["embedder"]["params"])
    else:
        embedder_cfg = load_config("embedder")
        embedder_params = embedder_cfg[emb_name]
    embedder = build_embedder(emb_name, embedder_params, nbits, mult)
    assert embedder_params.msg_processor.nbits == nbits and embedder_params.msg_processor.hidden_size == int(nbits * mult)     # embedder.py:258-259
    assert embedder.yuv == ("yuv" in emb_name)
    augmenter = Augmenter(augs={"identity": 1, "crop": 1}, augs_params={"crop": {"min_size": 0.5, "max_size": 1.0}}, masks={"kind": "none"}, num_augs=1)
    extractor_cfg = load_config("extractor")
    extractor_params = extractor_cfg[ext_name]
    extractor = build_extractor(ext_name, extractor_params, img_size, nbits)
    assert extractor_params.pixel_decoder.nbits == nbits                       # extractor.py:190
    if ext_name.startswith("convnext"):
        assert extractor_params.pixel_decoder.embed_dim == extractor_params.encoder.dims[-1]      # extractor.py:202
    if ext_name == "convnext_chunky":
        assert extractor_params.encoder.dims == [362, 724, 1448, 2896]         # extractor.py:192-197 at 1024 bits
    attenuation = None
    if att != "none":
        attenuation = JND(**load_config("attenuation")[att])
    wam = Videoseal(embedder, extractor, augmenter, attenuation, float(a["scaling_w"]), float(a.get("scaling_i", 1.0)), img_size=img_size,
                    chunk_size=8, step_size=4, blending_method="additive", lowres_attenuation=False)
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))[card]
    sd = wam.state_dict()
    assert list(sd) == list(ref)
    assert {k: list(v.shape) for k, v in sd.items()} == ref
    model_params = list(embedder.parameters()) + list(extractor.parameters())       # train.py:330
    assert all(p.requires_grad for p in model_params)
    n_named = len([k for k, _ in wam.named_parameters() if k.startswith(("embedder.", "detector."))])
    assert len(model_params) == n_named
    opt = torch.optim.AdamW(model_params, lr=1e-4)
    assert sum(len(g["params"]) for g in opt.param_groups) == len(model_params)
    # one engine configuration out of the two halves
    cfg = wam.embedder.cfg
    assert cfg is wam.detector.cfg and cfg.nbits == nbits and cfg.img_size == img_size and (cfg.chunk_size, cfg.step_size) == (8, 4)
    assert wam.blender.scaling_w == float(a["scaling_w"]) and (wam.chunk_size, wam.step_size, wam.lowres_attenuation) == (8, 4, False)
    # the merged configuration equals the card's (what videoseal.load builds), field by field where the constructor args agree
    from videoseal_amd.layout import cfg_from_card
    want = cfg_from_card(c)
    for f in ("nbits", "hidden", "yuv", "in_ch", "out_ch", "z", "mults", "num_blocks", "last_tanh", "depths", "dims", "stem_stride", "unet_act",
              "unet_norm", "extractor", "jnd_in", "jnd_out", "img_size"):
        assert getattr(cfg, f) == getattr(want, f), f
    if ext_name == "sam_small":
        for f in ("vit_dim", "vit_depth", "vit_heads", "vit_patch", "vit_window", "vit_global", "vit_out", "vit_mlp_ratio", "vit_rel_pos"):
            assert getattr(cfg, f) == getattr(want, f), f


def test_builders_accept_plain_dicts_and_refuse_what_is_outside_the_path():
    from videoseal_amd.builders import build_embedder, build_extractor
    cfg = {"msg_processor": {"nbits": 16, "hidden_size": 32, "msg_processor_type": "binary+concat"},
           "unet": {"in_channels": 1, "out_channels": 1, "z_channels": 8, "num_blocks": 2, "activation": "relu", "normalization": "batch",
                    "z_channels_mults": [1, 2], "last_tanh": True}}
    e = build_embedder("unet_tiny_yuv", cfg, 32)
    assert cfg["msg_processor"]["nbits"] == 32 and cfg["msg_processor"]["hidden_size"] == 64 and e.yuv and e.cfg.bott == 16 + 64
    x = build_extractor("convnext_tiny", {"encoder": {"depths": [1, 1, 1, 1], "dims": [8, 16, 24, 32]}, "pixel_decoder": {"upscale_stages": [1]}}, 64, 32)
    assert x.cfg.dims == [8, 16, 24, 32] and x.pixel_decoder.linear.weight.shape == (33, 32)
    for name in ("vae_small", "hidden", "patchmixer_x", "dvmark"):
        with pytest.raises(NotImplementedError):
            build_embedder(name, cfg, 32)
    for name in ("dino2s", "hidden", "dvmark", "nonsense"):
        with pytest.raises(NotImplementedError):
            build_extractor(name, cfg, 64, 32)
    with pytest.raises(NotImplementedError):
        build_extractor("convnext_tiny_pw", {"encoder": {"depths": [1, 1, 1, 1], "dims": [8, 16, 24, 32]},
                                             "pixel_decoder": {"upscale_stages": [4, 4, 2], "pixelwise": True}}, 64, 32)
    from videoseal_amd.model import Videoseal, get_dummy_augmenter
    with pytest.raises(ValueError, match="bits"):
        Videoseal(e, build_extractor("convnext_tiny", {"encoder": {"depths": [1, 1, 1, 1], "dims": [8, 16, 24, 32]}, "pixel_decoder": {}}, 64, 16),
                  get_dummy_augmenter())


def test_shim_is_closed_without_a_reference_checkout():
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("VIDEOSEAL_REFERENCE_ROOT", None)
    code = ("import videoseal, importlib\n"
            "for m in ('videoseal.losses', 'videoseal.utils.optim', 'videoseal.data'):\n"
            "    try:\n        importlib.import_module(m); raise SystemExit('resolved ' + m)\n    except ImportError: pass\n"
            "try:\n    videoseal.utils.bool_inst\n    raise SystemExit('attr')\nexcept AttributeError as e: assert 'VIDEOSEAL_REFERENCE_ROOT' in str(e)\n"
            "print('closed')")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "closed" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "videoseal")), reason="no reference checkout in this environment")
def test_shim_overlays_a_reference_checkout():
    """VIDEOSEAL_REFERENCE_ROOT: modules the shim does not define (train.py:55-72: videoseal.utils.dist / logger, utils.bool_inst, ...) come
    from the checkout; everything on the embed / extract path stays this package"""
    env = dict(os.environ, PYTHONPATH=ROOT, VIDEOSEAL_REFERENCE_ROOT=REF)
    code = ("import videoseal, videoseal_amd\n"
            "import videoseal.utils as utils\n"
            "import videoseal.utils.dist as udist\n"
            "import videoseal.utils.logger as ulogger\n"
            "from videoseal.models import Videoseal, Wam, build_embedder, build_extractor\n"
            "from videoseal.models.embedder import build_embedder as b2\n"
            "from videoseal.augmentation.valuemetric import JPEG\n"
            "from videoseal.modules.jnd import JND\n"
            "from videoseal.augmentation.augmenter import Augmenter\n"
            "from videoseal.evals.metrics import bit_accuracy, psnr\n"
            "assert udist.__file__.startswith('" + REF + "') and ulogger.__file__.startswith('" + REF + "')\n"
            "assert utils.bool_inst('yes') is True and callable(utils.get_sha)\n"
            "assert callable(udist.is_main_process) and hasattr(ulogger, 'MetricLogger')\n"
            "assert Videoseal is videoseal_amd.Videoseal and build_embedder is videoseal_amd.builders.build_embedder and b2 is build_embedder\n"
            "assert JPEG is videoseal_amd.augmentation.JPEG and JND is videoseal_amd.model.JND and Augmenter is videoseal_amd.augmentation.Augmenter\n"
            "assert bit_accuracy is videoseal_amd.metrics.bit_accuracy\n"
            "import importlib.util as iu\n"          # (importing these needs torchvision / lpips / timm, which the checkout's own code depends on)
            "for m in ('videoseal.modules.discriminator', 'videoseal.losses.videosealloss', 'videoseal.data.loader', 'videoseal.utils.optim',\n"
            "          'videoseal.utils.tensorboard', 'videoseal.utils.display', 'videoseal.models.baselines'):\n"
            "    assert iu.find_spec(m).origin.startswith('" + REF + "'), m\n"
            "for m in ('videoseal.models.videoseal', 'videoseal.models.embedder', 'videoseal.models.extractor', 'videoseal.modules.jnd',\n"
            "          'videoseal.augmentation.geometric', 'videoseal.evals.metrics', 'videoseal.utils.cfg'):\n"
            "    assert iu.find_spec(m).origin.startswith('" + ROOT + "'), m\n"
            "print('overlay ok')")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "overlay ok" in r.stdout, r.stdout + r.stderr


_CLOSURE = r'''
import ast, importlib, json, os, sys
ref = os.environ["VIDEOSEAL_REFERENCE_ROOT"]
files = {"train.py": None, "inference_streaming.py": None, "inference_av.py": None,
         "videoseal/evals/full.py": "videoseal.evals", "videoseal/evals/speed.py": "videoseal.evals"}
resolved, third_party, broken = [], {}, []
def third(e):
    while e is not None:
        if isinstance(e, ModuleNotFoundError) and e.name and not e.name.startswith("videoseal"):
            return e.name
        e = e.__cause__ or e.__context__
    return None
for rel, pkg in files.items():
    tree = ast.parse(open(os.path.join(ref, rel)).read())
    for node in ast.walk(tree):
        wants = []
        if isinstance(node, ast.Import):
            wants = [(a.name, None) for a in node.names if a.name.split(".")[0] == "videoseal"]
        elif isinstance(node, ast.ImportFrom):
            mod = node.module or ""
            if node.level:
                base = pkg.split(".")[: len(pkg.split(".")) - (node.level - 1)] if pkg else None
                if base is None:
                    continue
                mod = ".".join(base + ([mod] if mod else []))
            if mod.split(".")[0] == "videoseal":
                wants = [(mod, a.name) for a in node.names]
        for mod, name in wants:
            tag = f"{rel}:{node.lineno} {mod}" + (f".{name}" if name else "")
            try:
                m = importlib.import_module(mod)
                if name is not None and not hasattr(m, name):
                    importlib.import_module(mod + "." + name)          # `from pkg import submodule`
                resolved.append(tag)
            except BaseException as e:
                t = third(e)
                if t:
                    third_party.setdefault(t, []).append(tag)
                else:
                    broken.append(tag + " -> " + repr(e))
import videoseal_amd, videoseal.evals.metrics as M, videoseal.utils.cfg as C, videoseal.models as MD
own = {"Videoseal": MD.Videoseal is videoseal_amd.Videoseal, "bit_accuracy": M.bit_accuracy is videoseal_amd.metrics.bit_accuracy,
       "ssim": M.ssim is videoseal_amd.metrics.ssim, "accuracy": M.accuracy is videoseal_amd.metrics.accuracy,
       "setup_model_from_checkpoint": C.setup_model_from_checkpoint is videoseal_amd.cfg.setup_model_from_checkpoint,
       "setup_model": C.setup_model is videoseal_amd.cfg.setup_model}
print("CLOSURE " + json.dumps({"resolved": resolved, "third_party": third_party, "broken": broken, "own": own}))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "videoseal")), reason="no reference checkout in this environment")
def test_every_videoseal_import_of_the_reference_scripts_resolves_through_the_overlay():
    """import closure of the callers the shim claims to carry: every `import videoseal...` / `from videoseal... import name` statement of
    train.py, inference_streaming.py, inference_av.py, evals/full.py and evals/speed.py is parsed out of the checkout and resolved through
    the overlay.  A statement may fail only because a THIRD-PARTY package of the checkout's own code is absent here (printed); the path's
    names must be this package's objects."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, VIDEOSEAL_REFERENCE_ROOT=REF)
    r = subprocess.run([sys.executable, "-c", _CLOSURE], env=env, capture_output=True, text=True, cwd="/tmp")
    line = [l for l in r.stdout.splitlines() if l.startswith("CLOSURE ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(line[0][8:])
    print("third-party packages missing here:", {k: len(v) for k, v in res["third_party"].items()})
    assert not res["broken"], "\n".join(res["broken"])
    assert all(res["own"].values()), res["own"]
    must = ["train.py:65 videoseal.evals.metrics.accuracy", "train.py:65 videoseal.evals.metrics.iou", "train.py:65 videoseal.evals.metrics.ssim",
            "train.py:67 videoseal.models.build_extractor", "videoseal/evals/full.py:53 videoseal.utils.cfg.setup_model_from_checkpoint",
            "videoseal/evals/speed.py:34 videoseal.utils.cfg.setup_model_from_checkpoint", "videoseal/evals/full.py:46 videoseal.evals.metrics.msssim",
            "videoseal/evals/full.py:46 videoseal.evals.metrics.bd_rate", "videoseal/evals/full.py:46 videoseal.evals.metrics.vmaf_on_tensor",
            "inference_streaming.py:20 videoseal.evals.metrics.bit_accuracy"]
    missing = [m for m in must if m not in res["resolved"]]
    assert not missing, (missing, res["third_party"])
