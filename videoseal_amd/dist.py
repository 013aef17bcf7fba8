"""Multi-GPU sharding of the embed / extract path: one process per GPU, frames partitioned, one collective.

Frames are independent given the message; every `step_size`-aligned group of frames
reproduces the reference's key-frame positions, so a video is split into contiguous frame ranges whose
boundaries are multiples of `align` (16 = the streaming chunk of inference_streaming.py:180-184, itself a
multiple of step_size).  Embedding needs no collective.  Extraction ends with ONE exchange: an all-gather
of the per-frame bit logits (RCCL over xGMI on MI355X, `gloo` in the CPU tests), after which every rank
aggregates the full [F, k] matrix exactly like videoseal.py:411-428.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int, align: int = 16) -> Tuple[int, int]:
    """Contiguous [start, end) frame range of `rank`; all boundaries are multiples of `align`
    (the tail goes to the last non-empty rank).  Ranges cover [0, n_frames) without overlap."""
    if n_frames < 0 or world < 1 or not (0 <= rank < world) or align < 1:
        raise ValueError("bad shard arguments")
    units = (n_frames + align - 1) // align          # aligned groups (last one may be ragged)
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, n_frames), min(u1 * align, n_frames)


def all_shards(n_frames: int, world: int, align: int = 16) -> List[Tuple[int, int]]:
    return [shard_range(n_frames, r, world, align) for r in range(world)]


def gather_frame_logits(local: torch.Tensor, n_frames: int, align: int = 16, group=None) -> torch.Tensor:
    """all-gather of ragged per-rank [f_local, k] logits into the full [n_frames, k] matrix (frame order kept).
    Ranks pad to the largest shard so a single fixed-size all_gather is issued."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = all_shards(n_frames, world, align)
    a, b = shards[rank]
    if local.shape[0] != b - a:
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames, expected {b - a}")
    biggest = max(e - s for s, e in shards)
    k = local.shape[1]
    pad = torch.zeros(biggest, k, device=local.device, dtype=local.dtype)
    pad[: b - a] = local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # a host-side process group over device tensors (several ranks sharing ONE GPU, where RCCL refuses a second rank per device -- the
        # two-ranks-on-one-box runs of bench.py / tests; also a node without xGMI): the same single fixed-size all-gather through pinned
        # host staging buffers
        hin = torch.empty(biggest, k, dtype=local.dtype, pin_memory=True)
        hin.copy_(pad, non_blocking=True)
        torch.cuda.current_stream(local.device).synchronize()
        hout = torch.empty(world * biggest, k, dtype=local.dtype, pin_memory=True)
        dist.all_gather_into_tensor(hout, hin, group=group)
        out = hout.to(local.device, non_blocking=True)
    else:
        out = torch.empty(world * biggest, k, device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, pad, group=group)    # ONE fixed-size collective (RCCL ring over xGMI: 128 KiB per rank)
    if all(e - s == biggest for s, e in shards):
        return out
    return torch.cat([out[r * biggest: r * biggest + e - s] for r, (s, e) in enumerate(shards)], dim=0)


def extract_message_sharded(model, local_frames: torch.Tensor, n_frames: int, aggregation: str = "avg", align: int = 16,
                            interpolation: Optional[dict] = None, group=None) -> torch.Tensor:
    """Distributed `Videoseal.extract_message`: detect on the local shard, all-gather the logits, aggregate."""
    from .model import aggregate_bits
    it = interpolation or {"mode": "bilinear", "align_corners": False, "antialias": False}
    if local_frames.shape[0] > 0:
        local = model.detect(local_frames, is_video=True, interpolation=it)["preds"]
    else:
        local = torch.zeros(0, model.embedder.cfg.nbits + 1, device=local_frames.device)
    full = gather_frame_logits(local, n_frames, align, group)
    return (aggregate_bits(full[:, 1:], aggregation) > 0).squeeze().unsqueeze(0)


def check_alignment(model, align: int = 16) -> None:
    """Shard boundaries must not move key frames: every shard has to start on a key frame of the single-process run
    (align % step_size == 0), and in video_mode='interpolate' additionally on a chunk boundary, because the last key frame
    of a chunk is held instead of interpolated (videoseal.py:101-117) -- chunk_size*step_size frames per chunk."""
    step, chunk = int(model.step_size), int(model.chunk_size)
    if align % step != 0:
        raise ValueError(f"shard alignment {align} is not a multiple of step_size {step}: key frames would move")
    if getattr(model, "video_mode", "repeat") == "interpolate" and align % (step * chunk) != 0:
        raise ValueError(f"video_mode='interpolate' needs shards aligned to chunk_size*step_size = {step * chunk} frames (got {align})")


def embed_sharded(model, local_frames: torch.Tensor, msgs: torch.Tensor, align: int = 16, **kw) -> torch.Tensor:
    """Embedding of the local shard of a clip split with shard_range(..., align).  The result equals the corresponding slice of
    the single-process run (check_alignment enforces the conditions).  No collective: outputs stay sharded (or are written to
    disjoint file ranges)."""
    check_alignment(model, align)
    return model.embed(local_frames, msgs, is_video=True, **kw)["imgs_w"]


# ---- training under torch.distributed (train.py:437-448) ------------------------------------------------------------
def bn_all_reduce(group=None):
    """The exchange step of SyncBatchNorm as this library does it: every BatchNorm layer all-reduces ONE vector of 2*ld + 1 doubles --
    [sum x | sum x^2 | rows] of the local rows (csrc/net_ops.hip::vs_bn_partial_sums) -- and normalises with the statistics of the global
    batch (vs_bn_finish_sums).  torch's SyncBatchNorm all-gathers (mean, invstd, count) per rank and merges them; summing fp64 moments is
    the same statistic with one fixed-size collective and no per-rank merge kernel.  Ranks may hold different numbers of rows."""
    def reduce_(sums: torch.Tensor) -> None:
        if sums.dtype != torch.float64 or sums.dim() != 1:
            raise ValueError("the BatchNorm exchange vector is a 1-d float64 tensor")
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return reduce_


def convert_sync_batchnorm(model, group=None, reduce_=None):
    """`nn.SyncBatchNorm.convert_sync_batchnorm(wam)` of train.py:438-440 for a videoseal_amd model: from now on `model.train()` forwards
    normalise the U-Net's BatchNorm layers with the statistics of the batch of ALL ranks of `group` and update the running statistics with
    them (identical on every rank).  `reduce_` replaces the collective (tests).  Returns the model, like the torch call."""
    model._bn_sync = reduce_ if reduce_ is not None else bn_all_reduce(group)
    return model


def all_reduce_gradients(params, group=None, bucket_bytes: int = 64 << 20, average: bool = True) -> int:
    """The gradient exchange `DistributedDataParallel` performs for train.py:442-443, for gradients produced outside autograd
    (`training.DetectorStep` writes `.grad` directly): the gradients are packed, in parameter order, into flat fp32 buckets of at most
    `bucket_bytes`, every bucket is summed over the ranks with ONE all-reduce (all issued asynchronously before the first wait; RCCL rings
    over xGMI are per-link bound, so few large messages -- 64 MiB by default -- rather than one per tensor), divided by the world size
    (DDP averages) and scattered back into `.grad`.  Every rank must hold a gradient for the same parameters.  Returns the number of
    buckets.  Without an initialised process group this is a no-op (returns 0)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    plist = [p for p in params if p.requires_grad]
    missing = [i for i, p in enumerate(plist) if p.grad is None]
    if missing:
        raise ValueError(f"parameters {missing[:8]}{'...' if len(missing) > 8 else ''} have no gradient on this rank: the ranks would exchange "
                         f"different buckets")
    if bucket_bytes < 4:
        raise ValueError("bucket_bytes must hold at least one fp32 value")
    world = dist.get_world_size(group)
    buckets: List[List[torch.nn.Parameter]] = [[]]
    fill = 0
    for p in plist:                                     # parameter order = bucket order on every rank
        nbytes = p.grad.numel() * 4
        if buckets[-1] and fill + nbytes > bucket_bytes:
            buckets.append([])
            fill = 0
        buckets[-1].append(p)
        fill += nbytes
    buckets = [b for b in buckets if b]
    flats, works = [], []
    for b in buckets:
        flat = torch.cat([p.grad.detach().reshape(-1).float() for p in b])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        flats.append(flat)
    for b, flat, w in zip(buckets, flats, works):
        w.wait()
        if average:
            flat.div_(world)
        off = 0
        for p in b:
            n = p.grad.numel()
            p.grad.copy_(flat[off: off + n].view_as(p.grad))
            off += n
    return len(buckets)
