"""CPU-only checks of the host side: module surface / state_dict contract, card parsing, error behaviour,
and that the C-ABI library loads and exports every symbol include/videoseal_hip.h declares."""
import ctypes
import json
import os
import re

import pytest
import torch

import videoseal_amd
from videoseal_amd import native
from videoseal_amd.layout import cfg_from_card, load_card
from videoseal_amd.model import Videoseal, aggregate_bits
from tests._util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("card", ["videoseal_1.0", "pixelseal", "videoseal_0.0"])
def test_state_dict_contract(card):
    """keys, ORDER and shapes equal the reference's state_dict (dumped by tests/golden/make_golden.py)."""
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))[card]
    m = videoseal_amd.build(card)
    sd = m.state_dict()
    assert list(sd) == list(ref)
    assert {k: list(v.shape) for k, v in sd.items()} == ref
    # the message table is ONE tensor registered twice (embedder.py:141-142 + unet.py:128)
    assert sd["embedder.msg_processor.msg_embeddings.weight"].data_ptr() == sd["embedder.unet.msg_processor.msg_embeddings.weight"].data_ptr()
    # strict=False load of a reference-format checkpoint dict round-trips
    sd2 = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in sd.items()}
    msg = m.load_state_dict(sd2, strict=False)
    assert not msg.missing_keys and not msg.unexpected_keys
    assert torch.equal(m.embedder.unet.outc.weight, sd2["embedder.unet.outc.weight"])


def test_chunkyseal_cfg_and_keys():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["chunkyseal"]
    cfg = cfg_from_card(load_card(os.path.join(ROOT, "videoseal_amd", "cards", "chunkyseal.yaml")))
    assert cfg.dims == [362, 724, 1448, 2896] and cfg.depths == [3, 3, 27, 3] and cfg.stem_stride == 2
    assert cfg.hidden == 2048 and cfg.bott == 2560 and cfg.in_ch == 3 and cfg.out_ch == 3 and not cfg.yuv
    assert cfg.chunk_size == 32 and cfg.step_size == 8        # legacy videowam_* keys (cfg.py:112-118)
    from videoseal_amd.layout import detector_entries, embedder_entries
    mine = {"embedder." + n: list(s) for n, s, _ in embedder_entries(cfg)}
    mine.update({"detector." + n: list(s) for n, s, _ in detector_entries(cfg)})
    for k, v in mine.items():
        assert ref[k] == v, k
    assert len(ref) == len(mine) + 4                          # + rgb2yuv.M and the three JND kernels


def test_module_surface():
    m = videoseal_amd.build("videoseal_1.0")
    assert isinstance(m, Videoseal) and isinstance(m, torch.nn.Module)
    assert m.training                                          # load()/build() return train mode like the reference
    for name in ("embedder", "detector", "augmenter", "blender", "attenuation", "rgb2yuv"):
        assert isinstance(getattr(m, name), torch.nn.Module)
    assert (m.img_size, m.clamp, m.chunk_size, m.step_size, m.video_mode, m.lowres_attenuation) == (256, True, 32, 4, "repeat", False)
    assert m.blender.scaling_w == 0.2 and m.blender.scaling_i == 1.0 and m.embedder.yuv is True
    assert m.device.type == "cpu"
    msg = m.get_random_msg(3)
    assert msg.shape == (3, 256) and msg.dtype == torch.int64 and set(msg.unique().tolist()) <= {0, 1}
    rep = m.get_random_msg(2, nb_repetitions=4)
    assert torch.equal(rep[:, :64], rep[:, 64:128])
    assert m.embedder.get_last_layer() is m.embedder.unet.outc.weight
    n_emb = sum(p.numel() for p in m.embedder.parameters())
    n_ext = sum(p.numel() for p in m.detector.parameters())
    assert abs(n_emb / 1e6 - 23.66) < 0.05 and abs(n_ext / 1e6 - 33.37) < 0.05
    assert not m.attenuation.conv_lum.weight.requires_grad
    m.eval()
    assert not m.training and not m.embedder.training


def test_no_cpu_fallback_and_loud_errors():
    m = videoseal_amd.build("videoseal_1.0").eval()
    x = torch.rand(2, 3, 64, 64)
    with pytest.raises(native.NativeError, match="no CPU execution path"):
        m.embed(x, is_video=True)
    with pytest.raises(native.NativeError):
        m.detect(x)
    with pytest.raises(FileNotFoundError):
        videoseal_amd.load("no_such_card")
    with pytest.raises(TypeError):
        videoseal_amd.load(3)
    with pytest.raises(FileNotFoundError, match="Checkpoint"):
        videoseal_amd.load("videoseal")                        # card resolves, checkpoint is not available offline
    with pytest.raises(NotImplementedError):
        cfg_from_card({"args": {"nbits": 96, "img_size_proc": 256, "attenuation": "jnd_1_1"},
                       "embedder": {"model": "vae_small", "params": {}}, "extractor": {"model": "sam_small", "params": {}}})


def test_aggregation_variants():
    p = torch.tensor([[1.0, -2.0], [3.0, 0.5], [-1.0, 0.25]])
    assert torch.allclose(aggregate_bits(p, "avg"), p.mean(0))
    assert torch.allclose(aggregate_bits(p, "squared_avg"), (p * p.abs()).mean(0))
    assert torch.allclose(aggregate_bits(p, "l1norm_avg"), (p * p.abs().sum(1, keepdim=True)).mean(0))
    assert torch.allclose(aggregate_bits(p, "l2norm_avg"), (p * p.norm(dim=1, keepdim=True)).mean(0))
    assert aggregate_bits(p, None) is p
    with pytest.raises(ValueError):
        aggregate_bits(p, "median")


def test_f16x2_operand_split_of_the_weights():
    """engine.split_f16x2 (the weight side of the default 2 x f16 arithmetic, conv_common.h Arith<2>): w * w_mul = hi + lo with w_mul a
    power of two that puts max|w| into [2^13, 2^14); |w * w_mul - hi - lo| <= 2^-23 |w * w_mul| for every weight whose low term is a
    normal f16 and <= 2^-25 (half a denormal quantum) otherwise; blocked LDS image = a permutation of the planes."""
    import math
    import torch
    from videoseal_amd.engine import pack_blocked, split_bf16x3, split_f16x2
    g = torch.Generator().manual_seed(5)
    for scale in (1e-4, 0.02, 1.0, 300.0):
        w = torch.randn(70, 9 * 32, generator=g) * scale
        w[0, 0] = 0.0
        planes, w_mul = split_f16x2(w)
        assert planes.shape == (2, 70, 288) and planes.dtype == torch.int16
        assert math.log2(w_mul) == round(math.log2(w_mul))
        ws = w.double() * w_mul
        assert 2 ** 13 <= ws.abs().max() < 2 ** 14
        hi, lo = planes[0].view(torch.float16).double(), planes[1].view(torch.float16).double()
        err = (ws - hi - lo).abs()
        normal = (ws - hi).abs() >= 2.0 ** -14
        assert (err[normal] <= 2.0 ** -23 * ws.abs()[normal]).all()
        assert (err[~normal] <= 2.0 ** -25 + 1e-30).all()
        blk = pack_blocked(planes, 9)
        assert blk.shape == (3, 18, 2, 64, 8) and sorted(blk.reshape(-1).tolist()) == sorted(planes.reshape(-1).tolist() + [0] * (blk.numel() - planes.numel()))
        # the exact 3 x bf16 split of the other arithmetic: three truncated terms that add up to w exactly
        p3 = split_bf16x3(w)
        terms = [(p3[i].to(torch.int32) << 16).view(torch.float32).double() for i in range(3)]
        assert torch.equal(terms[0] + terms[1] + terms[2], w.double())


def test_c_abi_exports_every_declared_symbol():
    """the shared library loads on a machine without a GPU and exports exactly what the header declares."""
    header = open(os.path.join(ROOT, "include", "videoseal_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no prototypes found"
    lib = native.lib()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/videoseal_hip.h but not exported"
    assert sorted(native.EXPORTS) == declared
    assert lib.vs_version() == 3 and lib.vs_arch() == b"gfx950"
    assert b"bad argument" in lib.vs_error_string(-1)
    # argument validation happens on the host before any launch: null pointers are rejected without a GPU
    assert lib.vs_layernorm_act(None, 4, 8, 8, None, None, 1e-6, 0, None, 8, None) == -1
    assert lib.vs_conv_gemm(None, None) == -1
    assert lib.vs_sizeof_conv_desc() == ctypes.sizeof(native.ConvDesc) and lib.vs_sizeof_tail_desc() == ctypes.sizeof(native.TailDesc)


def test_metrics_match_oracle():
    from oracle import videoseal_ref as R
    from videoseal_amd.metrics import bit_accuracy, psnr
    a = torch.rand(3, 3, 16, 16)
    b = (a + 0.02 * torch.randn_like(a)).clamp(0, 1)
    assert torch.allclose(psnr(a, b), R.psnr(a, b)) and torch.allclose(psnr(a, b, True), R.psnr(a, b, True))
    p = torch.randn(4, 32)
    t = torch.randint(0, 2, (4, 32))
    assert torch.equal(bit_accuracy(p, t), R.bit_accuracy(p, t))
    pix = torch.randn(2, 8, 5, 5)
    assert bit_accuracy(pix, torch.randint(0, 2, (2, 8))).shape == (2,)


def _reference_metrics_module():
    """the checkout's own evals/metrics.py, executed as a file (its `pytorch_msssim` import is absent here: a dummy module stands in, so
    only the functions that do not touch it are usable)"""
    import importlib.util
    import sys
    import types
    path = "/root/reference/videoseal/evals/metrics.py"
    if not os.path.isfile(path):
        pytest.skip("no reference checkout in this environment")
    added = "pytorch_msssim" not in sys.modules and importlib.util.find_spec("pytorch_msssim") is None
    if added:
        sys.modules["pytorch_msssim"] = types.ModuleType("pytorch_msssim")
    try:
        spec = importlib.util.spec_from_file_location("_ref_metrics_for_test", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if added:
            sys.modules.pop("pytorch_msssim", None)
    return mod


def test_mask_and_message_metrics_equal_the_reference_functions():
    """train.py:65,649,667-670 / evals/full.py:46: accuracy, iou, linf, pvalue, capacity, bit_accuracy(+_1msg, masked) against the
    unmodified evals/metrics.py on seeded inputs incl. empty masks / empty unions"""
    R = _reference_metrics_module()
    from videoseal_amd import metrics as M
    g = torch.Generator().manual_seed(5)
    preds = torch.randn(5, 1, 12, 9, generator=g)
    targets = (torch.rand(5, 1, 12, 9, generator=g) > 0.6).float()
    targets[3] = 0
    preds[3] = -1.0                                                      # empty union for label 1
    for thr in (0.0, 0.3):
        assert torch.equal(M.accuracy(preds, targets, thr), R.accuracy(preds, targets, thr))
        for label in (0, 1):
            assert torch.equal(M.iou(preds.clone(), targets, thr, label), R.iou(preds.clone(), targets, thr, label))
    a, b = torch.rand(2, 3, 20, 20, generator=g), torch.rand(2, 3, 20, 20, generator=g)
    assert torch.equal(M.linf(a, b), R.linf(a, b)) and torch.equal(M.psnr(a, b), R.psnr(a, b)) and torch.equal(M.psnr(a, b, True), R.psnr(a, b, True))
    bits = torch.randint(0, 2, (4, 32), generator=g)
    logits = (bits.float() * 2 - 1) * torch.rand(4, 32, generator=g) + 0.4 * torch.randn(4, 32, generator=g)
    logits[1] = bits[1].float() * 2 - 1                                  # a perfect row: p log p at 0
    assert torch.equal(M.bit_accuracy(logits, bits), R.bit_accuracy(logits, bits))
    assert torch.equal(M.pvalue(logits, bits), R.pvalue(logits, bits))
    assert torch.equal(M.capacity(logits, bits), R.capacity(logits, bits))
    pix = torch.randn(3, 8, 6, 6, generator=g)
    tb = torch.randint(0, 2, (3, 8), generator=g)
    mask = (torch.rand(3, 1, 6, 6, generator=g) > 0.5).float()
    mask[:] = mask[0]                                                    # (masked_select + view needs equal counts, as in the reference)
    assert torch.equal(M.bit_accuracy(pix, tb, mask), R.bit_accuracy(pix, tb, mask))
    assert torch.equal(M.bit_accuracy_1msg(pix, tb), R.bit_accuracy_1msg(pix, tb))
    assert torch.equal(M.bit_accuracy_1msg(pix, tb, mask), R.bit_accuracy_1msg(pix, tb, mask))


def test_ssim_and_msssim_against_an_independent_float64_evaluation():
    """pytorch_msssim is absent (parity unpinned against it): the torch restatement is checked against a scipy.ndimage float64 evaluation of
    the published definition -- 11-tap Gaussian (sigma 1.5), valid region, K = (0.01, 0.03); MS-SSIM with 2x2 mean pooling"""
    import numpy as np
    from scipy.ndimage import correlate1d
    from videoseal_amd.metrics import msssim, ssim
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 176, 163, generator=g)
    y = (x + 0.05 * torch.randn(2, 3, 176, 163, generator=g)).clamp(0, 1)
    c = np.arange(11) - 5.0
    w = np.exp(-c ** 2 / (2 * 1.5 ** 2))
    w /= w.sum()

    def blur(a):
        a = correlate1d(correlate1d(a, w, axis=-2, mode="constant"), w, axis=-1, mode="constant")
        return a[..., 5:-5, 5:-5]

    def parts(a, b):
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m1, m2 = blur(a), blur(b)
        s1, s2, s12 = blur(a * a) - m1 * m1, blur(b * b) - m2 * m2, blur(a * b) - m1 * m2
        cs = (2 * s12 + c2) / (s1 + s2 + c2)
        return (((2 * m1 * m2 + c1) / (m1 * m1 + m2 * m2 + c1)) * cs).mean((-2, -1)), cs.mean((-2, -1))

    a, b = x.double().numpy(), y.double().numpy()
    s_ref, _ = parts(a, b)
    assert np.abs(ssim(x, y).numpy() - s_ref.mean(1)).max() < 2e-5
    assert torch.allclose(ssim(x, x), torch.ones(2), atol=1e-6)
    weights = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]
    acc = np.ones((2, 3))
    for lv, wt in enumerate(weights):
        s, cs = parts(a, b)
        acc *= np.maximum(s if lv == 4 else cs, 0) ** wt
        if lv < 4:
            pad = [(0, 0), (0, 0), (a.shape[2] % 2, a.shape[2] % 2), (a.shape[3] % 2, a.shape[3] % 2)]     # avg_pool2d(padding=s % 2): zeros, counted
            a, b = (np.pad(v, pad)[:, :, : (v.shape[2] + 2 * pad[2][0]) // 2 * 2, : (v.shape[3] + 2 * pad[3][0]) // 2 * 2] for v in (a, b))
            a, b = (v.reshape(v.shape[0], v.shape[1], v.shape[2] // 2, 2, v.shape[3] // 2, 2).mean((3, 5)) for v in (a, b))
    assert np.abs(msssim(x, y).numpy() - acc.mean(1)).max() < 5e-5
    with pytest.raises(AssertionError):
        msssim(x[..., :160, :], y[..., :160, :])


def test_setup_model_from_checkpoint_reads_a_training_checkpoint(tmp_path):
    """evals/full.py:53,313 -> utils/cfg.py:52-179: a `.pth` as train.py writes it ({'model', 'args'}; args name the config YAMLs and use the
    pre-rename `videowam_*` keys) -> config -> Videoseal with the weights loaded; a card name goes to the card loader, a baseline raises, a
    missing file is FileNotFoundError"""
    from videoseal_amd import cfg as C
    args = {"embedder_config": "configs/embedder.yaml", "extractor_config": "configs/extractor.yaml", "attenuation_config": "configs/attenuation.yaml",
            "embedder_model": "unet_small2_yuv_quant", "extractor_model": None, "attenuation": "jnd_1_1", "nbits": 64, "hidden_size_multiplier": 1,
            "img_size_proc": 256, "scaling_w": 0.3, "scaling_i": 1.0, "videowam_chunk_size": 16, "videowam_step_size": 2}
    probe = C.setup_model.__wrapped__ if hasattr(C.setup_model, "__wrapped__") else None  # noqa: F841
    ck = tmp_path / "checkpoint.pth"
    torch.save({"model": {}, "args": args}, ck)
    config = C.get_config_from_checkpoint(ck)
    assert isinstance(config, C.VideosealConfig) and config.embedder.model == "unet_small2_yuv_quant"
    ext_default = videoseal_amd.builders.load_config("extractor")["model"]
    assert config.extractor.model == ext_default
    m0 = C.setup_model(config, ck)                                       # strict=False: an empty state_dict loads
    sd = {k: (torch.randn_like(v) if v.is_floating_point() else v) for k, v in m0.state_dict().items()}
    torch.save({"model": sd, "args": args}, ck)
    m = C.setup_model_from_checkpoint(str(ck))
    assert isinstance(m, Videoseal) and m.chunk_size == 16 and m.step_size == 2 and m.img_size == 256
    assert m.embedder.cfg.nbits == 64 and float(m.blender.scaling_w) == pytest.approx(0.3)
    assert torch.equal(m.state_dict()["embedder.unet.outc.weight"], sd["embedder.unet.outc.weight"])
    assert config.embedder.params["msg_processor"]["nbits"] == 64        # the factories write back into the config, embedder.py:258-259
    # train.py:562 writes the args as a YAML string (`omegaconf.OmegaConf.to_yaml(params)`), utils/cfg.py:63-64 parses it back
    import yaml
    ck_str = tmp_path / "checkpoint_yaml_args.pth"
    torch.save({"model": sd, "args": yaml.safe_dump(args)}, ck_str)
    cfg_str = C.get_config_from_checkpoint(ck_str)
    assert cfg_str.embedder.model == "unet_small2_yuv_quant" and cfg_str.args["nbits"] == 64
    ms = C.setup_model_from_checkpoint(str(ck_str))
    assert isinstance(ms, Videoseal) and ms.chunk_size == 16 and ms.embedder.cfg.nbits == 64
    assert torch.equal(ms.state_dict()["embedder.unet.outc.weight"], sd["embedder.unet.outc.weight"])
    torch.save({"model": sd, "args": "just a scalar"}, ck_str)
    with pytest.raises(Exception, match="params dictionary"):
        C.get_config_from_checkpoint(ck_str)
    with pytest.raises(NotImplementedError, match="baseline"):
        C.setup_model_from_checkpoint("baseline/hidden")
    with pytest.raises(FileNotFoundError):
        C.setup_model_from_checkpoint("no_such_card_anywhere")
    with pytest.raises(FileNotFoundError):
        C.setup_model(C.get_config_from_checkpoint(ck), tmp_path / "missing.pth")
    from videoseal.utils.cfg import setup_model_from_checkpoint as shim_fn
    assert shim_fn is C.setup_model_from_checkpoint


def test_videoseal_import_shim_resolves_the_reference_paths():
    """inference_streaming.py:18-20, inference_av.py:20: `import videoseal` / `videoseal.models` / `videoseal.evals.metrics` / `utils.cfg`"""
    import videoseal
    import videoseal_amd
    from videoseal.augmentation import Identity, JPEG, get_validation_augs  # noqa: F401
    from videoseal.augmentation.augmenter import Augmenter, get_dummy_augmenter  # noqa: F401
    from videoseal.augmentation.sequential import Sequential  # noqa: F401
    from videoseal.evals.metrics import bit_accuracy, psnr
    from videoseal.models import Videoseal, Wam
    from videoseal.models.videoseal import Videoseal as V2
    from videoseal.modules.jnd import JND  # noqa: F401
    from videoseal.utils.cfg import setup_model_from_model_card
    assert videoseal.load is videoseal_amd.load and setup_model_from_model_card is videoseal_amd.load
    assert Videoseal is videoseal_amd.Videoseal and V2 is Videoseal and issubclass(Videoseal, Wam)
    assert bit_accuracy is videoseal_amd.metrics.bit_accuracy and psnr is videoseal_amd.metrics.psnr
    m = videoseal_amd.build("videoseal_1.0")
    assert isinstance(m, Videoseal) and m.training and type(m.augmenter).__name__ == "Augmenter"
    with pytest.raises(FileNotFoundError):
        videoseal.load("videoseal")          # card found, checkpoint absent (no network): the reference's error type


def test_detector_step_has_no_cpu_path():
    """videoseal_amd.training.DetectorStep (train.py:517-523): loud on a CPU model, for the ConvNeXt-V2 extractor and for the ViT extractor of
    the legacy card alike (no CPU fallback of the training path either)"""
    from videoseal_amd.training import DetectorStep
    for card in ("videoseal_1.0", "videoseal_0.0"):
        m = videoseal_amd.build(card).train()
        with pytest.raises(native.NativeError, match="no CPU execution path"):
            DetectorStep(m).step(torch.rand(2, 3, 256, 256), torch.zeros(2, m.embedder.cfg.nbits))


def test_validation_tables_match_the_reference():
    """augmentation/__init__.py:12-130 row for row (classes + fixed strengths), incl. the codec rows of the video tables
    (fixture: tests/golden/make_golden_tables.py run on the unmodified reference)"""
    import json
    import os
    from videoseal_amd import augmentation as A
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "validation_tables.json")) as f:
        gold = json.load(f)

    def rows(table):
        out = []
        for aug, params in table:
            name = ("Sequential(" + ",".join(t.__class__.__name__ for t in aug.transforms) + ")") if isinstance(aug, A.Sequential) else aug.__class__.__name__
            out.append([name, [list(p) if isinstance(p, tuple) else p for p in params]])
        return out
    assert rows(A.get_validation_augs(False)) == gold["validation_image"]
    assert rows(A.get_validation_augs(True)) == gold["validation_video"]
    assert rows(A.get_validation_augs(False, only_identity=True)) == gold["identity"]
    assert rows(A.get_validation_augs(False, only_combined=True)) == gold["combined_image"]
    assert rows(A.get_validation_augs(True, only_combined=True)) == gold["combined_video"]
    assert rows(A.get_validation_augs_subset(False)) == gold["subset_image"]
    assert rows(A.get_validation_augs_subset(True)) == gold["subset_video"]
    assert repr(A.H264(20, 30)) == "H264proxy" and "backend=proxy" in repr(A.VideoCompression())      # the stand-in is visible in tables and logs
    assert A.H264(20, 30).aug_name == "H264" and A.H264(20, 30).backend_name == "H264proxy"            # `selected_aug` stays the reference's class name


def test_result_buffers_of_a_cpu_caller():
    """videoseal_amd/model.py::_result_buffer / _to_caller: shape / dtype / device of what a CPU caller gets back, pass-through of
    non-tensors and of tensors already on the caller's device, pageable memory on request (no GPU needed for any of it)."""
    import videoseal_amd.model as M
    t = M._result_buffer((2, 3, 4), torch.uint8, "cpu")
    assert t.shape == (2, 3, 4) and t.dtype == torch.uint8 and t.device.type == "cpu"
    x = torch.arange(6.0)
    assert M._to_caller(x, "cpu") is x and M._to_caller(None, "cpu") is None and M._to_caller("crop_0.5", "cpu") == "crop_0.5"
    old = M._PINNED_RESULTS
    try:
        M._PINNED_RESULTS = False
        assert not M._result_buffer((4,), torch.float32, torch.device("cpu")).is_pinned()
    finally:
        M._PINNED_RESULTS = old


def test_bench_gpus_n_without_a_launcher_refuses_cleanly_when_the_node_has_too_few_gpus():
    """`python bench.py --gpus 2` (no torch.distributed.run around it): bench.py launches itself, or -- fewer GPUs than asked for, here none --
    prints ONE JSON error line and exits with code 2 instead of dying with a launcher message"""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could run it")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stderr[-500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    err = json.loads(lines[0])
    assert err["n_gpus_requested"] == 2 and err["n_gpus_visible"] == torch.cuda.device_count() and "error" in err


def test_videoseal_lib_selects_another_build_of_the_library():
    """native.LIB_PATH: VIDEOSEAL_LIB names another build of the same ABI (tools/ab_libs.sh, same-box A/B of two source trees); unset, the
    in-tree library is the one that loads -- checked in a child interpreter because the path is fixed at import"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "from videoseal_amd import native; print(native.LIB_PATH)"
    env = dict(os.environ, VIDEOSEAL_LIB="/somewhere/else/libvideoseal_hip.so")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip().endswith("/somewhere/else/libvideoseal_hip.so"), out.stderr
    env.pop("VIDEOSEAL_LIB")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == os.path.join(root, "videoseal_amd", "csrc", "libvideoseal_hip.so"), out.stderr


def test_bench_cpu_leg_times_the_unmodified_reference_when_a_checkout_is_present():
    """bench.py's cpu_baseline leg (`--cpu-baseline-only`: no GPU needed): kind 'reference' = the unmodified module (stub-import recipe) wherever a
    checkout is reachable, kind 'port' = the oracle otherwise (VS_BENCH_CPU_PORT=1 forces it; the GPU box has no checkout)"""
    import subprocess
    import sys
    have_ref = os.path.isdir("/root/reference/videoseal")
    for force_port in (False, True):
        env = dict(os.environ, **({"VS_BENCH_CPU_PORT": "1"} if force_port else {}))
        env.pop("VIDEOSEAL_REFERENCE_ROOT", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--size", "128"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["kind"] == ("reference" if (have_ref and not force_port) else "port") and d["value"] > 0 and d["cores"] >= 1 and d["unit"] == "frames/s"
        assert ("UNMODIFIED reference" in d["sample"]) == (d["kind"] == "reference")
