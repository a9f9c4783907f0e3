import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import parity_common as pc
from lip2speech_amd import synth
from oracle import l2s_oracle as orc
B, T = int(os.environ.get("B", 2)), int(os.environ.get("T", 9))
sd = synth.synth_state_dict()
video = synth.synth_video(B, T, tag="enc-bn-train")
torch.manual_seed(11)
cot = torch.randn(B, T, 768, dtype=torch.float64)
enc = [k for k in sd if k.startswith("encoder.")]
is_stat = lambda k: k.endswith(("running_mean", "running_var"))
par = [k for k in enc if sd[k].is_floating_point() and not is_stat(k)]
res = {}
for dt in (torch.float64, torch.float32):
    sdx = {k: (sd[k].detach().clone().to(dt).requires_grad_(k in par) if sd[k].is_floating_point() else sd[k]) for k in enc}
    with orc.batch_statistics():
        f = orc.encoder_forward(sdx, video.to(dt))
    (f * cot.to(dt)).sum().backward()
    res[dt] = {k: sdx[k].grad.double() for k in par}
nm = pc.native_model(sd)
params = {k: sd[k].clone().cuda() for k in enc if sd[k].is_floating_point()}
grads = {k: torch.zeros_like(params[k]) for k in par}
nm.train_bind(params, grads)
nm.train_set_bn(True, 0.1)
_, feat, tape = nm.train_encoder_fwd(video.cuda())
dvis = torch.zeros(B, T, 1024, device="cuda"); dvis[:, :, :768] = cot.float().cuda()
nm.train_encoder_bwd(video.cuda(), dvis, tape)
print("feat err", pc.maxdiff(feat, f.double()))
for k in par:
    r64, r32 = res[torch.float64][k], res[torch.float32][k]
    g = grads[k].cpu().double().reshape(r64.shape)
    sc = max(r64.abs().max().item(), 1e-6)
    e64 = (g - r64).abs().max().item() / sc; e32 = (g - r32).abs().max().item() / sc; e3264 = (r32 - r64).abs().max().item() / sc
    n64 = ((g - r64).norm() / max(r64.norm(), 1e-9)).item(); n3264 = ((r32 - r64).norm() / max(r64.norm(), 1e-9)).item()
    if max(e64, e32) > 2e-3:
        print(f"{k:42s} hip-64 {e64:.1e} hip-32 {e32:.1e} 32-64 {e3264:.1e} | L2rel hip-64 {n64:.1e} 32-64 {n3264:.1e}")
