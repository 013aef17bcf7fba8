import torch, sys
sys.path.insert(0, '/root/repo')
import videoseal_amd
from torch.profiler import profile, ProfilerActivity
model = videoseal_amd.build("videoseal_1.0", seed=0).eval().cuda()
model.chunk_size = 32
x = torch.rand(32, 3, 768, 768, device="cuda")
msgs = torch.randint(0, 2, (32, 256))
for _ in range(3):
    w = model.embed(x, msgs, is_video=False)["imgs_w"]; p = model.detect(w, is_video=True)["preds"]
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    w = model.embed(x, msgs, is_video=False)["imgs_w"]; p = model.detect(w, is_video=True)["preds"]
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
