"""The RCCL branch on one GPU (world size 1 over the `nccl` backend = RCCL): the sharded extraction equals the single-process
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


def test_sharded_extraction_over_rccl_world1():
    import torch.distributed as dist
    from videoseal_amd.dist import embed_sharded, extract_message_sharded, gather_frame_logits, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
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
    finally:
        dist.destroy_process_group()


def test_bench_distributed_step_on_one_gpu():
    env = dict(os.environ, VS_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4", "--size", "256",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "all-gather" in line["config"]["workload"]
