"""One rank of tests/test_gpu_zdist.py::test_ddp_two_ranks_on_one_gpu_match_the_single_process_step: the inner loop of train.py:626-643 on this
rank's half of the batch, model wrapped exactly as train.py:438-446 does (SyncBatchNorm conversion, DistributedDataParallel).  Both ranks
share cuda:0 (RCCL refuses two ranks on one device, so the process group is gloo, which stages device tensors through the host); launched
as `python -m tests._ddp_worker <out.pt>` with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment."""
import os
import sys

import torch
import torch.distributed as dist


def batch_and_model():
    from oracle.weights import make_state_dict, tiny_spec
    from oracle.inputs import synthetic_frames, synthetic_msgs
    from tests.test_gpu_e2e import make_model
    import videoseal_amd.augmentation as G
    spec = tiny_spec()
    model = make_model(spec, make_state_dict(spec, seed=3))
    model.augmenter = G.Augmenter(masks={"kind": "none"}, augs={"identity": 1}, augs_params={}, num_augs=1)
    model.train()
    imgs = synthetic_frames(4, 64, 80, seed=17).cuda()
    msgs = synthetic_msgs(4, spec.nbits, seed=17)
    return model, imgs, msgs


LOSS_KW = dict(percep_loss="mse", percep_weight=1.0, detect_weight=0.0, decode_weight=1.0, balanced=False)


def step(fwd, model, imgs, msgs):
    from oracle import loss as OL
    masks = torch.ones(imgs.shape[0], 1, imgs.shape[2], imgs.shape[3], device=imgs.device)
    out = fwd(imgs, masks, msgs, is_video=False)
    loss, _ = OL.videoseal_loss(imgs, out["imgs_w"], out["masks"], out["msgs"].cuda(), out["preds"], last_layer=None, **LOSS_KW)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}


def extract_job(rank: int, world: int, out_path: str) -> None:
    """frames sharded over the ranks (contiguous 16-aligned ranges), embed locally, ONE all-gather of the logits (videoseal_amd/dist.py)"""
    from oracle.inputs import synthetic_frames, synthetic_msgs
    from oracle.weights import make_state_dict, tiny_spec
    from tests.test_gpu_e2e import make_model
    from videoseal_amd.dist import embed_sharded, extract_message_sharded, gather_frame_logits, shard_range
    spec = tiny_spec()
    model = make_model(spec, make_state_dict(spec, seed=3))
    model.chunk_size, model.step_size = 4, 2
    frames = synthetic_frames(40, 80, 96, seed=70)
    msgs = synthetic_msgs(1, spec.nbits, seed=70)
    a, b = shard_range(40, rank, world, 16)
    w = embed_sharded(model, frames[a:b].cuda(), msgs, align=16)
    logits = gather_frame_logits(model.detect(w, is_video=True)["preds"], 40, 16)
    bits = {agg: extract_message_sharded(model, w, 40, aggregation=agg).cpu() for agg in ("avg", "squared_avg", "l1norm_avg", "l2norm_avg")}
    torch.cuda.synchronize()
    torch.save({"range": (a, b), "imgs_w": w.cpu(), "logits": logits.cpu(), "bits": bits, "device": torch.cuda.current_device()}, out_path)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    backend = os.environ.get("VS_DDP_BACKEND", "gloo")           # "nccl" = RCCL, one device per rank (needs >= world GPUs)
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        probe = torch.ones(3, device="cuda", dtype=torch.float64)
        dist.all_reduce(probe)                                   # gloo with device tensors: not every build has it
        assert float(probe[0]) == world
    except Exception as e:       # noqa: BLE001
        torch.save({"unsupported": repr(e)}, out_path)
        return
    if os.environ.get("VS_DDP_JOB") == "extract":
        extract_job(rank, world, out_path)
        dist.barrier()
        dist.destroy_process_group()
        return
    from videoseal_amd.dist import convert_sync_batchnorm
    model, imgs, msgs = batch_and_model()
    convert_sync_batchnorm(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev])
    n = imgs.shape[0] // world
    loss, grads = step(ddp, model, imgs[rank * n:(rank + 1) * n], msgs[rank * n:(rank + 1) * n])
    bn = torch.cat([b.detach().double().flatten().cpu() for k, b in model.named_buffers() if "running" in k])
    torch.save({"loss": loss, "grads": grads, "bn": bn}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
