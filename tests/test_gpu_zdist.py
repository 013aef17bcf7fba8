"""(Named *_zdist so that it is collected LAST: the process-group tests run after every kernel / parity test of the suite.)
The RCCL branch on one GPU (world size 1 over the `nccl` backend = RCCL): the sharded extraction equals the single-process
call, and bench.py's distributed step (barrier, all-gather of the logits, max-over-ranks timing) runs end to end."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.inputs import synthetic_frames, synthetic_msgs  # noqa: E402
from oracle.weights import make_state_dict, tiny_spec  # noqa: E402
from tests.test_gpu_e2e import make_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.fixture(scope="module")
def rccl_world1():
    """ONE world-size-1 process group over the `nccl` (= RCCL) backend for the tests of this file (initialised once per process)"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_sharded_extraction_over_rccl_world1(rccl_world1):
    from videoseal_amd.dist import embed_sharded, extract_message_sharded, gather_frame_logits, shard_range
    spec = tiny_spec()
    model = make_model(spec, make_state_dict(spec, seed=3))
    model.chunk_size, model.step_size = 4, 2
    frames = synthetic_frames(40, 80, 96, seed=70).cuda()
    msgs = synthetic_msgs(1, spec.nbits, seed=70)
    a, b = shard_range(40, 0, 1, 16)
    assert (a, b) == (0, 40)
    w = embed_sharded(model, frames[a:b], msgs, align=16)
    assert torch.equal(w, model.embed(frames, msgs, is_video=True)["imgs_w"])
    single = model.extract_message(w)
    for agg in ("avg", "squared_avg", "l1norm_avg", "l2norm_avg"):
        assert torch.equal(extract_message_sharded(model, w, 40, aggregation=agg), model.extract_message(w, aggregation=agg))
    assert torch.equal(extract_message_sharded(model, w, 40), single)
    logits = model.detect(w, is_video=True)["preds"]
    assert torch.equal(gather_frame_logits(logits, 40, 16), logits)         # one all_gather_into_tensor on the RCCL communicator


def test_bench_distributed_step_on_one_gpu():
    env = dict(os.environ, VS_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4", "--size", "256",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "all-gather" in line["config"]["workload"]


# ---- SyncBatchNorm (train.py:438-440): BatchNorm statistics of the global batch, one all-reduce of fp64 moments per layer
def test_bn_partial_sums_add_up_to_the_batch_statistics():
    """vs_bn_partial_sums on two ragged halves, added, then vs_bn_finish_sums == vs_bn_batch_stats on the whole tensor; on one piece the
    pair is bit-identical to it (same summation order)."""
    from videoseal_amd import native as N
    L = N.lib()
    st = N.stream()
    g = torch.Generator().manual_seed(3)
    rows, C, ld = 5000, 20, 24
    x = torch.zeros(rows, ld)
    x[:, :C] = torch.randn(rows, C, generator=g) * 2 + 0.5
    x = x.cuda()
    gamma, beta = torch.rand(ld, generator=g).cuda() + 0.5, torch.randn(ld, generator=g).cuda()

    def fresh():
        return torch.zeros(ld).cuda(), torch.ones(ld).cuda(), torch.empty(ld).cuda(), torch.empty(ld).cuda()

    def part(t):
        return torch.empty(int(L.vs_bn_partial_doubles(t.shape[0], ld)), dtype=torch.float64, device="cuda")
    rm0, rv0, sc0, sh0 = fresh()
    N.check(L.vs_bn_batch_stats(N.ptr(x), rows, C, ld, N.ptr(gamma), N.ptr(beta), 1e-5, 0.1, N.ptr(rm0), N.ptr(rv0), N.ptr(part(x)),
                                N.ptr(sc0), N.ptr(sh0), st), "vs_bn_batch_stats")
    # one piece: bit-identical
    sums = torch.empty(2 * ld + 1, dtype=torch.float64, device="cuda")
    rm1, rv1, sc1, sh1 = fresh()
    N.check(L.vs_bn_partial_sums(N.ptr(x), rows, C, ld, N.ptr(part(x)), N.ptr(sums), st), "vs_bn_partial_sums")
    assert float(sums[-1]) == rows
    N.check(L.vs_bn_finish_sums(N.ptr(sums), C, ld, N.ptr(gamma), N.ptr(beta), 1e-5, 0.1, N.ptr(rm1), N.ptr(rv1), N.ptr(sc1), N.ptr(sh1), st),
            "vs_bn_finish_sums")
    for a, b in ((rm0, rm1), (rv0, rv1), (sc0, sc1), (sh0, sh1)):
        assert torch.equal(a[:C], b[:C])
    # two ragged pieces (3000 + 2000 rows), moments added as the all-reduce does
    xa, xb = x[:3000].contiguous(), x[3000:].contiguous()
    sa, sb = torch.empty_like(sums), torch.empty_like(sums)
    N.check(L.vs_bn_partial_sums(N.ptr(xa), 3000, C, ld, N.ptr(part(xa)), N.ptr(sa), st), "vs_bn_partial_sums")
    N.check(L.vs_bn_partial_sums(N.ptr(xb), 2000, C, ld, N.ptr(part(xb)), N.ptr(sb), st), "vs_bn_partial_sums")
    tot = sa + sb
    assert float(tot[-1]) == rows and ((tot[:C] - sums[:C]).abs() <= 1e-12 * sums[:C].abs() + 1e-9).all()
    rm2, rv2, sc2, sh2 = fresh()
    N.check(L.vs_bn_finish_sums(N.ptr(tot), C, ld, N.ptr(gamma), N.ptr(beta), 1e-5, 0.1, N.ptr(rm2), N.ptr(rv2), N.ptr(sc2), N.ptr(sh2), st),
            "vs_bn_finish_sums")
    for a, b in ((rm0, rm2), (rv0, rv2), (sc0, sc2), (sh0, sh2)):
        assert (a[:C] - b[:C]).abs().max() <= 3e-7 * a[:C].abs().max()
    # and against torch's own batch statistics
    xc = x[:, :C].double()
    mean, var = xc.mean(0), xc.var(0, unbiased=False)
    assert (sc2[:C].double() - gamma[:C].double() / torch.sqrt(var + 1e-5)).abs().max() < 1e-6
    assert (rm2[:C].double() - 0.1 * mean).abs().max() < 1e-6
    assert (rv2[:C].double() - (0.9 + 0.1 * xc.var(0, unbiased=True))).abs().max() < 1e-5


def _train_forward(model, imgs, msgs):
    from videoseal_amd import augmentation as G
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs={"identity": 1}, augs_params={}, num_augs=1)
    model.train()
    return model(imgs, torch.ones(imgs.shape[0], 1, *imgs.shape[-2:], device=imgs.device), msgs, is_video=False)


def _bn_buffers(model):
    sd = model.state_dict()
    return torch.cat([v.flatten().float() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))]).cpu()


def test_sync_batchnorm_world1_over_rccl_is_bit_identical(rccl_world1):
    from videoseal_amd.dist import convert_sync_batchnorm
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    imgs, msgs = synthetic_frames(4, 72, 88, seed=31).cuda(), synthetic_msgs(4, spec.nbits, seed=31)
    model = make_model(spec, sd)              # ONE instance for both runs: the same tile choices, so the comparison can be bit-exact
    ref = {k: v.clone() for k, v in _train_forward(model, imgs, msgs).items() if k in ("imgs_w", "preds_w")}
    ref_bn = _bn_buffers(model)
    model.load_state_dict(sd, strict=True)    # running statistics back to their initial values
    convert_sync_batchnorm(model)
    calls = []
    inner = model._bn_sync
    model._bn_sync = lambda s: (calls.append(s.numel()), inner(s))[1]          # fp64 all-reduce on the RCCL communicator
    out = _train_forward(model, imgs, msgs)
    torch.cuda.synchronize()
    assert len(calls) > 0 and all(n % 2 == 1 for n in calls)             # one exchange of 2*ld + 1 doubles per BatchNorm layer
    assert torch.equal(out["imgs_w"], ref["imgs_w"]) and torch.equal(out["preds_w"], ref["preds_w"])
    assert torch.equal(_bn_buffers(model), ref_bn)


def test_sync_batchnorm_two_ranks_reproduce_the_global_batch():
    """Two model replicas on this GPU, each holding a ragged half of the batch (3 + 1 frames), run in lock step on two host threads; the
    exchange adds their fp64 moment vectors (what the all-reduce does).  Every rank must produce its slice of the single-process result on
    the whole batch -- which tests/test_gpu_fwd.py pins to the reference's train-mode forward -- and identical running statistics."""
    import threading
    from videoseal_amd.dist import convert_sync_batchnorm
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    imgs, msgs = synthetic_frames(4, 72, 88, seed=31).cuda(), synthetic_msgs(4, spec.nbits, seed=31)
    plain = make_model(spec, sd)
    ref = {k: v.clone() for k, v in _train_forward(plain, imgs, msgs).items() if k in ("imgs_w", "preds_w")}
    world, cuts = 2, [(0, 3), (3, 4)]
    bar = threading.Barrier(world, timeout=120)
    slots = [None] * world

    def reducer(rank):
        def reduce_(s):
            slots[rank] = s
            bar.wait()
            tot = slots[0] + slots[1]          # same order on every rank; all launches share the default stream
            bar.wait()
            s.copy_(tot)
        return reduce_
    models = [convert_sync_batchnorm(make_model(spec, sd), reduce_=reducer(r)) for r in range(world)]
    outs, errs = [None] * world, []

    def run(rank):
        try:
            a, b = cuts[rank]
            outs[rank] = _train_forward(models[rank], imgs[a:b], msgs[a:b])
        except BaseException as e:       # noqa: BLE001 -- reported below; a broken barrier releases the other thread
            errs.append((rank, repr(e)))
            bar.abort()
    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errs and all(o is not None for o in outs), errs
    torch.cuda.synchronize()
    for (a, b), o in zip(cuts, outs):
        assert (o["preds_w"] - ref["preds_w"][a:b]).abs().max() <= 1e-5 * ref["preds_w"].abs().max()     # (tile choices may differ with the batch size)
        assert (o["imgs_w"] - ref["imgs_w"][a:b]).abs().max() <= 1e-5
    b0, b1, bp = _bn_buffers(models[0]), _bn_buffers(models[1]), _bn_buffers(plain)
    assert torch.equal(b0, b1)
    assert (b0 - bp).abs().max() <= 1e-5 * bp.abs().max()
    local_only = _train_forward(make_model(spec, sd), imgs[:3], msgs[:3])           # without the exchange the result differs
    assert (local_only["preds_w"] - ref["preds_w"][:3]).abs().max() > 1e-4 * ref["preds_w"].abs().max()


# ---- train.py:442-446: nn.parallel.DistributedDataParallel around the module, gradients through DDP's autograd hooks
def test_ddp_wrapped_training_step_over_rccl_world1(rccl_world1):
    """DDP(model) + the unmodified inner loop: the parameters enter the HIP graph nodes as inputs, so DDP's AccumulateGrad hooks fire and its
    reducer all-reduces the buckets over RCCL (the detector's buckets while the embedder's backward is still running).  World size 1: the
    gradients must equal the un-wrapped step's; the adaptive-weight probes (torch.autograd.grad with retain_graph) must not trip the reducer."""
    from oracle import loss as OL
    from tests._util import load_golden
    from tests.test_gpu_train import _setup
    from videoseal_amd.dist import convert_sync_batchnorm
    spec = tiny_spec()
    sd = make_state_dict(spec, seed=3)
    meta = load_golden("tiny_bwd_img_balanced")["meta"]

    def run(wrap):
        model, imgs, msgs, masks = _setup(spec, sd, meta)
        fwd = model
        if wrap:
            convert_sync_batchnorm(model)                   # train.py:438-440 (here: the moment all-reduce of dist.py over the same communicator)
            fwd = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
        torch.manual_seed(meta["torch_seed"])
        out = fwd(imgs.cuda(), masks.cuda(), msgs, is_video=False)
        out["preds"] /= meta["temperature"]
        loss, log = OL.videoseal_loss(imgs.cuda(), out["imgs_w"], out["masks"], out["msgs"].cuda(), out["preds"],
                                      last_layer=model.embedder.get_last_layer(), **meta["loss_kw"])
        loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, float(loss), {k: float(v) for k, v in log.items()}
    g0, l0, log0 = run(False)
    g1, l1, log1 = run(True)
    assert abs(l0 - l1) < 1e-6 and all(abs(log0[k] - log1[k]) <= 1e-5 * max(1.0, abs(log0[k])) for k in log0)
    assert set(g0) == set(g1)
    for k in g0:
        assert (g0[k] - g1[k]).abs().max() <= 1e-5 * float(g0[k].abs().max()) + 1e-12, k


def test_ddp_two_ranks_on_one_gpu_match_the_single_process_step(tmp_path):
    """World size 2 for real: two processes, each with half of the batch, SyncBatchNorm conversion + DistributedDataParallel as in
    train.py:438-446 (tests/_ddp_worker.py).  Both ranks must end with the gradients of the single-process step on the whole batch (DDP
    averages; the loss terms are batch means) and with the whole batch's BatchNorm running statistics."""
    import os
    import subprocess
    import sys
    from tests import _ddp_worker as W
    model, imgs, msgs = W.batch_and_model()
    loss_ref, g_ref = W.step(model, model, imgs, msgs)
    bn_ref = torch.cat([b.detach().double().flatten().cpu() for k, b in model.named_buffers() if "running" in k])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-m", "tests._ddp_worker", outs[r]], cwd=root, env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0].decode(errors="replace")[-3000:])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("timeout")
    assert all(p.returncode == 0 for p in procs), logs
    res = [torch.load(o) for o in outs]
    if any("unsupported" in r for r in res):
        pytest.skip(f"gloo cannot reduce device tensors in this build: {[r.get('unsupported') for r in res]}")
    assert (res[0]["bn"] - res[1]["bn"]).abs().max() == 0
    assert (res[0]["bn"] - bn_ref).abs().max() <= 1e-5 * bn_ref.abs().max()
    assert abs(0.5 * (res[0]["loss"] + res[1]["loss"]) - loss_ref) <= 1e-5 * abs(loss_ref)
    assert set(res[0]["grads"]) == set(g_ref) == set(res[1]["grads"])
    for k in g_ref:
        a, b, r = res[0]["grads"][k], res[1]["grads"][k], g_ref[k]
        assert torch.equal(a, b), k                                             # both ranks hold the averaged gradient
        assert (a - r).abs().max() <= 1e-4 * float(r.abs().max()) + 1e-10, (k, float((a - r).abs().max()), float(r.abs().max()))


# ------------------------------------------------------------------------------------------------ two DEVICES over RCCL (needs a >= 2-GPU box)
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


def _run_two_ranks(tmp_path, job):
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               VS_DDP_BACKEND="nccl", VS_DDP_JOB=job)
    outs = [str(tmp_path / f"{job}_rank{r}.pt") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-m", "tests._ddp_worker", outs[r]], cwd=ROOT, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=900)[0].decode(errors="replace")[-3000:])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("timeout")
    assert all(p.returncode == 0 for p in procs), logs
    res = [torch.load(o) for o in outs]
    assert not any("unsupported" in r for r in res), res
    return res


@two_gpus
def test_sharded_extraction_on_two_devices_over_rccl(tmp_path):
    """videoseal_amd/dist.py for real: rank r embeds and detects its contiguous 16-aligned frame range on cuda:r, ONE RCCL all-gather of the
    logits, every aggregation of extract_message -- equal to the single-process call on the whole clip (frames are independent)"""
    res = _run_two_ranks(tmp_path, "extract")
    spec = tiny_spec()
    model = make_model(spec, make_state_dict(spec, seed=3))
    model.chunk_size, model.step_size = 4, 2
    frames = synthetic_frames(40, 80, 96, seed=70).cuda()
    msgs = synthetic_msgs(1, spec.nbits, seed=70)
    w = model.embed(frames, msgs, is_video=True)["imgs_w"]
    logits = model.detect(w, is_video=True)["preds"].cpu()
    assert [r["range"] for r in res] == [(0, 32), (32, 40)] and [r["device"] for r in res] == [0, 1]
    assert torch.equal(torch.cat([r["imgs_w"] for r in res]), w.cpu())
    for r in res:           # shard boundaries are chunk boundaries (16 | 8 frames per embed chunk, 4 per detect batch): bit-identical to the whole clip
        assert torch.equal(r["logits"], logits)
        for agg, bits in r["bits"].items():
            assert torch.equal(bits, model.extract_message(w, aggregation=agg).cpu()), agg


@two_gpus
def test_ddp_on_two_devices_over_rccl_matches_the_single_process_step(tmp_path):
    """train.py:438-446 on two devices: SyncBatchNorm conversion + DistributedDataParallel over RCCL, half of the batch per rank -> both ranks
    end with the single-process gradients of the whole batch and its BatchNorm running statistics"""
    from tests import _ddp_worker as W
    model, imgs, msgs = W.batch_and_model()
    loss_ref, g_ref = W.step(model, model, imgs, msgs)
    bn_ref = torch.cat([b.detach().double().flatten().cpu() for k, b in model.named_buffers() if "running" in k])
    res = _run_two_ranks(tmp_path, "ddp")
    assert (res[0]["bn"] - res[1]["bn"]).abs().max() == 0
    assert (res[0]["bn"] - bn_ref).abs().max() <= 1e-5 * bn_ref.abs().max()
    assert abs(0.5 * (res[0]["loss"] + res[1]["loss"]) - loss_ref) <= 1e-5 * abs(loss_ref)
    for k in g_ref:
        a, b, r = res[0]["grads"][k], res[1]["grads"][k], g_ref[k]
        assert torch.equal(a, b), k
        assert (a - r).abs().max() <= 1e-4 * float(r.abs().max()) + 1e-10, (k, float((a - r).abs().max()), float(r.abs().max()))


@two_gpus
def test_bench_launches_itself_on_two_gpus():
    """`python bench.py --gpus 2` without a launcher re-executes under torch.distributed.run and prints ONE line with n_gpus 2"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--size", "256",
                        "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["allgather_ms"] > 0 and line["scaling"] == "weak"


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="the refusal path of a 1-GPU box")
def test_bench_refuses_more_gpus_than_the_node_has():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2
    err = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert err["n_gpus_requested"] == 2 and err["n_gpus_visible"] == torch.cuda.device_count() and "error" in err


# ---- two ranks SHARING one GPU through a host-side process group (VS_BENCH_COLLECTIVE=gloo): the multi-rank code path of bench.py and dist.py
# ---- executed on the hardware the build has, before the driver's first 8-GPU run (RCCL refuses two ranks on one device)
def _bench_two_ranks_one_gpu(extra, tmp_path, timeout=1500, ranks=2):
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VS_BENCH_COLLECTIVE"] = "gloo"
    env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", "0").split(",")[0]        # ONE device for both ranks
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(ranks)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]                 # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_stream_leg_with_two_ranks_on_one_gpu_gathers_the_single_process_logits(tmp_path):
    """`bench.py --gpus 2 --mode stream --frames 64`, two ranks on ONE device, logits exchanged through gloo: rank count seen by the communicator,
    16-aligned contiguous shards, strong scaling, and the gathered [64, 1 + nbits] logits equal what one process computes for the two shards
    (rank r's frames are synthetic_batch(seed 1000 + r); chunks are independent, so the sharded run must reproduce them bit for bit)"""
    dump = str(tmp_path / "preds.pt")
    line = _bench_two_ranks_one_gpu(["--mode", "stream", "--frames", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra",
                                     "--no-kernel-timers", "--dump-preds", dump], tmp_path)
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["scaling"] == "strong" and line["allgather_ms"] > 0
    assert line["shards"] == [[0, 32], [32, 64]] and "gloo" in line["collective"] and line["value"] > 0
    got = torch.load(dump)
    assert got.shape == (64, 257)
    sys.path.insert(0, ROOT)
    import bench
    import videoseal_amd
    from videoseal_amd.streaming import embed_detect_chunks
    model = videoseal_amd.build("videoseal_1.0", seed=0).eval().cuda()
    model.chunk_size = max(model.chunk_size, 32)
    msgs = torch.randint(0, 2, (1, 256), generator=torch.Generator().manual_seed(5))
    ref = torch.cat([embed_detect_chunks(model, bench.synthetic_batch(32, 768, torch.device("cuda"), seed=1000 + r), msgs, chunk=16, lowres_attenuation=True,
                                         overlap=True).cpu() for r in range(2)])
    assert torch.equal(got, ref), float((got - ref).abs().max())


@pytest.mark.gpu
def test_bench_default_legs_with_two_ranks_on_one_gpu(tmp_path):
    """the DEFAULT command of the driver's scaling run (`bench.py --gpus 2 --steps K --warmup W`: image-mode headline, weak scaling, plus the
    stream_1024 and ChunkySeal legs that run at every rank count) with two ranks on one device: exits cleanly with one line"""
    line = _bench_two_ranks_one_gpu(["--steps", "2", "--warmup", "1", "--no-kernel-timers"], tmp_path)
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["shards"] == [[0, 32], [32, 64]] and line["config"]["batch_per_gpu"] == 32
    legs = line["configs"]
    st = [v for k, v in legs.items() if k.startswith("stream_1024")][0]
    ck = [v for k, v in legs.items() if k.startswith("chunkyseal")][0]
    assert "error" not in st and st["n_gpus"] == 2 and st["scaling"] == "strong" and st["value"] > 0 and st["allgather_ms"] > 0
    assert "error" not in ck and ck["n_gpus"] == 2 and ck["value"] > 0
    assert not any(k.startswith(("video_step4", "chain", "train_step")) for k in legs)           # single-rank legs stay out of a multi-rank line


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu_preflight(tmp_path):
    """the driver's 8-GPU command (`bench.py --gpus 8 --steps K --warmup W`, one rank per GPU under torch.distributed.run) with EIGHT Python ranks
    on the one device the build has (gloo group, VS_BENCH_COLLECTIVE): all eight build the model, take the shards [[0, 128], ..., [896, 1024]] of the
    1024-frame clip, gather, and exit with ONE line; the ChunkySeal leg builds 1.8 B random parameters in every rank.  The line carries the
    host side of a rank (peak RSS, start-up seconds; max over ranks) -- recorded in DESIGN.md section 8.  (videoseal/utils/dist.py:210-213)"""
    line = _bench_two_ranks_one_gpu(["--steps", "1", "--warmup", "1", "--no-kernel-timers"], tmp_path, timeout=2400, ranks=8)
    assert line["n_gpus"] == 8 and line["n_ranks_seen"] == 8 and line["scaling"] == "weak"
    assert line["shards"] == [[32 * r, 32 * r + 32] for r in range(8)] and line["allgather_ms"] > 0
    st = [v for k, v in line["configs"].items() if k.startswith("stream_1024")][0]
    ck = [v for k, v in line["configs"].items() if k.startswith("chunkyseal")][0]
    assert "error" not in st and st["n_gpus"] == 8 and st["scaling"] == "strong" and st["value"] > 0
    assert "error" not in ck and ck["n_gpus"] == 8 and ck["value"] > 0
    host = line["host"]
    print("8-rank preflight:", json.dumps({"host": host, "host_after_chunkyseal": ck.get("host"), "allgather_ms": line["allgather_ms"], "stream_allgather_ms": st["allgather_ms"],
                                           "image_ms_per_step": line["ms_per_step"], "stream_ms_per_step": st["ms_per_step"]}))
    assert host["rss_gb_max_over_ranks"] < 64 and host["startup_s_max_over_ranks"] < 900
    assert ck["host"]["rss_gb_max_over_ranks"] < 64            # 8 ranks x (1.8 B random fp32 parameters + their packed images in flight)
    out = os.environ.get("VS_PREFLIGHT_OUT")
    if out:
        with open(out, "w") as f:
            json.dump(line, f)
