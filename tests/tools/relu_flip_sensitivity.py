"""How far does the U-Net gradient move when the forward differs in the last bits?  CPU only (oracle): autograd through the VideoSeal 1.0 U-Net
in fp32 against fp64, and fp32 with two thread counts.  The net is piecewise linear in its ReLUs, so rounding differences flip a few ReLU
decisions and shift the gradient discretely: this is the noise floor of any gradient comparison that does not share the ReLU decisions
(tests/test_gpu_bwd_unet.py shares them).  Measured here: worst 1.3 %, median 0.27 % of a tensor's largest gradient element (fp32 vs fp64)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import videoseal_ref as R
from oracle.inputs import synthetic_frames, synthetic_msgs
from oracle.weights import make_state_dict, spec_from_card
from tests.test_oracle_golden import CARDS
torch.set_num_threads(8)
spec = spec_from_card(os.path.join(CARDS, "videoseal_1.0.yaml")); sd = make_state_dict(spec, seed=0); n=2; seed=52
S = spec.img_size
imgs = synthetic_frames(n, S, S, seed=seed)
x01 = (0.299 * imgs[:, 0:1] + 0.587 * imgs[:, 1:2] + 0.114 * imgs[:, 2:3]).contiguous()
msgs = synthetic_msgs(n, spec.nbits, seed=seed)
dd = torch.randn(n, spec.out_ch, S, S, generator=torch.Generator().manual_seed(seed)) * 1e-3
names = [k for k, v in sd.items() if k.startswith("embedder.unet.") and v.dtype.is_floating_point and "running" not in k]
def run(dt, threads=8):
    torch.set_num_threads(threads)
    sdg = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    for k in names: sdg[k].requires_grad_(True)
    ref = R.embedder_forward(sdg, spec, x01.to(dt), msgs, {})
    ref.backward(dd.to(dt))
    return {k: sdg[k].grad for k in names if sdg[k].grad is not None}
g32, g64 = run(torch.float32), run(torch.float64)
errs = sorted([(float((g32[k].double()-g64[k]).abs().max()/g64[k].abs().max()), k) for k in g64], reverse=True)
print("vs10 fp32 vs fp64 (max-rel):", [(round(e,5),k) for e,k in errs[:6]], "median", errs[len(errs)//2][0])
g32b = run(torch.float32, threads=3)
errs = sorted([(float((g32[k]-g32b[k]).abs().max()/g32[k].abs().max()), k) for k in g32], reverse=True)
print("vs10 fp32 8 threads vs 3 threads:", [(round(e,5),k) for e,k in errs[:4]], "median", errs[len(errs)//2][0])
