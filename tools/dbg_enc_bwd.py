import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import parity_common as pc
from lip2speech_amd import synth
from oracle import l2s_oracle as orc
B, T = 1, 29
sd = synth.synth_state_dict()
video = synth.synth_video(B, T, tag=f"enc-train{T}")
torch.manual_seed(T)
cot = torch.randn(B, T, 768, dtype=torch.float64)
is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked"))
enc = [k for k in sd if k.startswith("encoder.")]
par = [k for k in enc if sd[k].is_floating_point() and not is_buf(k)]
res = {}
for dt in (torch.float64, torch.float32):
    sdx = {k: (sd[k].to(dt).requires_grad_(k in par) if sd[k].is_floating_point() else sd[k]) for k in enc}
    f = orc.encoder_forward(sdx, video.to(dt))
    (f * cot.to(dt)).sum().backward()
    res[dt] = {k: sdx[k].grad.double() for k in par}
nm = pc.native_model(sd)
params = {k: sd[k].cuda() for k in par}
grads = {k: torch.zeros_like(v) for k, v in params.items()}
nm.train_bind(params, grads)
_, feat, tape = nm.train_encoder_fwd(video.cuda())
dvis = torch.zeros(B, T, 1024, device="cuda"); dvis[:, :, :768] = cot.float().cuda()
nm.train_encoder_bwd(video.cuda(), dvis, tape)
for k in par:
    r64, r32 = res[torch.float64][k], res[torch.float32][k]
    g = grads[k].cpu().double().reshape(r64.shape)
    sc = r64.abs().max().item()
    e64 = (g - r64).abs().max().item() / sc; e32 = (g - r32).abs().max().item() / sc; e3264 = (r32 - r64).abs().max().item() / sc
    if max(e64, e32, e3264) > 1e-3:
        d = (g - r64).abs().flatten()
        print(f"{k:45s} hip-vs-64 {e64:.2e} hip-vs-32 {e32:.2e} 32-vs-64 {e3264:.2e}  n>{0.5*d.max():.1e}: {(d > 0.5*d.max()).sum().item()} of {d.numel()}")
