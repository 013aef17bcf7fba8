"""N > 1 path on CPU: frame sharding + the single logits all-gather, world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from videoseal_amd.dist import all_shards, gather_frame_logits, shard_range
from videoseal_amd.model import aggregate_bits


@pytest.mark.parametrize("n,world,align", [(1024, 8, 16), (100, 3, 16), (5, 4, 16), (0, 2, 16), (33, 2, 4), (128, 1, 16)])
def test_shards_cover_without_overlap(n, world, align):
    sh = all_shards(n, world, align)
    assert sh[0][0] == 0 and sh[-1][1] == n
    for (a, b), (c, d) in zip(sh, sh[1:]):
        assert b == c and a <= b
    for a, b in sh:
        assert a % align == 0 or a == n          # every shard starts on a key-frame boundary
    assert sum(b - a for a, b in sh) == n
    if n == 1024 and world == 8:
        assert sh == [(i * 128, (i + 1) * 128) for i in range(8)]      # BASELINE config 4


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_frames, k, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    full = torch.randn(n_frames, k, generator=g)            # what a single process would have computed
    a, b = shard_range(n_frames, rank, world, 16)
    gathered = gather_frame_logits(full[a:b].clone(), n_frames, 16)
    ok = torch.equal(gathered, full)
    msg = (aggregate_bits(gathered, "avg") > 0)
    q.put((rank, ok, msg.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [64, 50])
def test_gather_frame_logits_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, 12, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    full = torch.randn(n_frames, 12, generator=g)
    want = (full.mean(0) > 0).tolist()
    for rank, ok, msg in res:
        assert ok, f"rank {rank}: gathered logits differ from the single-process matrix"
        assert msg == want                                   # every rank decodes the same message


def test_shard_alignment_rules():
    """shard starts must be key frames; interpolate mode additionally needs chunk-aligned shards"""
    from types import SimpleNamespace
    from videoseal_amd.dist import check_alignment
    check_alignment(SimpleNamespace(step_size=4, chunk_size=32, video_mode="repeat"), 16)
    check_alignment(SimpleNamespace(step_size=4, chunk_size=4, video_mode="interpolate"), 16)
    with pytest.raises(ValueError, match="step_size"):
        check_alignment(SimpleNamespace(step_size=3, chunk_size=8, video_mode="repeat"), 16)
    with pytest.raises(ValueError, match="interpolate"):
        check_alignment(SimpleNamespace(step_size=4, chunk_size=32, video_mode="interpolate"), 16)


# ---- SyncBatchNorm exchange (train.py:438-440): one all-reduce of [sum x | sum x^2 | rows] per BatchNorm layer
def _bn_worker(rank, world, port, q):
    from videoseal_amd.dist import bn_all_reduce
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(11, 8, generator=g).double() * 3 + 1       # the global batch: 11 rows of 8 channels, ragged split 7 + 4
    mine = x[:7] if rank == 0 else x[7:]
    sums = torch.cat([mine.sum(0), (mine * mine).sum(0), torch.tensor([float(mine.shape[0])], dtype=torch.float64)])
    bn_all_reduce()(sums)
    n = float(sums[-1])
    mean, var = sums[:8] / n, sums[8:16] / n - (sums[:8] / n) ** 2
    q.put((rank, n, float((mean - x.mean(0)).abs().max()), float((var - x.var(0, unbiased=False)).abs().max())))
    dist.destroy_process_group()


def test_sync_batchnorm_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, dm, dv in res:
        assert n == 11.0 and dm < 1e-12 and dv < 1e-12, (rank, n, dm, dv)      # every rank ends with the GLOBAL batch statistics


def test_sync_batchnorm_conversion_is_host_state_only():
    from types import SimpleNamespace
    from videoseal_amd.dist import bn_all_reduce, convert_sync_batchnorm
    m = SimpleNamespace(_bn_sync=None)
    assert convert_sync_batchnorm(m) is m and callable(m._bn_sync)
    v = torch.arange(5, dtype=torch.float64)
    m._bn_sync(v)                                                # no process group: the local statistics are the global ones
    assert torch.equal(v, torch.arange(5, dtype=torch.float64))
    with pytest.raises(ValueError, match="float64"):
        bn_all_reduce()(torch.zeros(5))
    marker = lambda s: None                                      # noqa: E731
    assert convert_sync_batchnorm(m, reduce_=marker)._bn_sync is marker


# ---- gradient exchange of distributed training (train.py:442-443: DistributedDataParallel) for gradients written outside autograd
def _grad_worker(rank, world, port, bucket_bytes, q):
    from videoseal_amd.dist import all_reduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(11)
    shapes = [(3, 5), (7,), (2, 3, 4), (1,), (64, 9)]
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    base = [torch.randn(*s, generator=g) for s in shapes]
    for p, b in zip(params, base):
        p.grad = b * (rank + 1)                       # rank r holds (r + 1) * base: the mean over 2 ranks is 1.5 * base
    frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)        # frozen parameters (the embedder) take no part
    nb = all_reduce_gradients(params + [frozen], bucket_bytes=bucket_bytes)
    err = max(float((p.grad - 1.5 * b).abs().max()) for p, b in zip(params, base))
    params[1].grad = None
    try:
        all_reduce_gradients(params)
        raised = False
    except ValueError:
        raised = True
    q.put((rank, nb, err, raised, frozen.grad is None))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,want_buckets", [(64 << 20, 1), (100, 3), (4, 5)])
def test_gradient_all_reduce_world2(bucket_bytes, want_buckets):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nb, err, raised, frozen_untouched in res:
        assert nb == want_buckets and err < 1e-6 and raised and frozen_untouched, (rank, nb, err, raised)


def test_gradient_all_reduce_without_a_process_group_is_a_no_op():
    from videoseal_amd.dist import all_reduce_gradients
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    assert all_reduce_gradients([p]) == 0 and torch.equal(p.grad, torch.ones(3))
