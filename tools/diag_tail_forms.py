import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from videoseal_amd import native as N
from videoseal_amd.native import TailDesc
L = N.lib()
g = torch.Generator(device="cuda").manual_seed(11)
F_, H, W, S = 2, 64, 541, 32
x = torch.rand(F_, 3, H, W, device="cuda", generator=g).contiguous()
taps = (C.c_float * 43)(*([1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 2, 0, 2, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1] + [-1, 0, 1, -2, 0, 2, -1, 0, 1] + [1, 2, 1, 0, 0, 0, -1, -2, -1]))
for Cd, att in ((1, 0), (3, 0), (1, 1)):
    delta = (0.3 * torch.randn(F_, Cd, S, S, device="cuda", generator=g)).contiguous()
    def run(variant):
        out = torch.full_like(x, -7.0); pw = torch.full((F_, Cd, H, W), -7.0, device="cuda")
        d = TailDesc()
        d.imgs, d.out, d.preds_w = N.ptr(x), N.ptr(out), N.ptr(pw)
        d.delta, d.hmap_lowres, d.taps43 = N.ptr(delta), None, C.cast(taps, C.c_void_p)
        d.F, d.H, d.W, d.S_h, d.S_w, d.Cd = F_, H, W, S, S, Cd
        d.step, d.video_mode, d.total_key = 1, 0, F_
        d.attenuate, d.clamp, d.antialias = att, 1, 1
        d.scaling_i, d.scaling_w, d.io_u8, d.variant = 1.0, 0.2, 0, variant
        N.check(L.vs_embed_tail(C.byref(d), N.stream()), "t"); torch.cuda.synchronize()
        return out, pw
    a, pa = run(2); b, pb = run(4)
    ne = (a != b); npw = (pa != pb)
    print("Cd", Cd, "att", att, "out differs:", int(ne.sum()), "of", ne.numel(), " preds_w differs:", int(npw.sum()), "max", float((pa - pb).abs().max()))
    if npw.any():
        idx = npw.nonzero()[:8].tolist(); print("  first preds_w diffs", idx)
    elif ne.any():
        idx = ne.nonzero()[:8].tolist(); print("  first out diffs", idx, [ (float(a[tuple(i)]), float(b[tuple(i)])) for i in ne.nonzero()[:3].tolist()])
