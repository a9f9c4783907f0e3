"""Central-difference check of the HIP encoder backward against the HIP encoder forward itself (batch-statistics BatchNorm)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
B, T = 2, 9
sd = synth.synth_state_dict()
enc = {k: v for k, v in sd.items() if k.startswith("encoder.")}
native.set_option("refresh_map", 1)
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in enc.items()}, list(enc.keys()))
native.set_option("refresh_map", 0)
is_stat = lambda k: k.endswith(("running_mean", "running_var"))
par = [k for k in enc if enc[k].is_floating_point() and not is_stat(k)]
params = {k: enc[k].clone().cuda() for k in enc if enc[k].is_floating_point()}
grads = {k: torch.zeros_like(params[k]) for k in par}
nm.train_bind(params, grads)
bn = int(os.environ.get("BN", 1))
nm.train_set_bn(bool(bn), 0.1)
video = synth.synth_video(B, T, tag="enc-bn-train").cuda()
torch.manual_seed(11)
cot = torch.randn(B, T, 768, device="cuda")
def loss():
    nm.train_refresh_weights()
    _, feat, tape = nm.train_encoder_fwd(video)
    return float((feat.double() * cot.double()).sum()), tape
L0, tape = loss()
dvis = torch.zeros(B, T, 1024, device="cuda"); dvis[:, :, :768] = cot
nm.train_encoder_bwd(video, dvis, tape)
g = {k: grads[k].clone() for k in par}
base = {k: params[k].clone() for k in par}
for trial in range(4):
    torch.manual_seed(100 + trial)
    d = {k: torch.randn_like(base[k]) * base[k].pow(2).mean().sqrt() for k in par}
    if trial == 3:      # along the gradient itself
        d = {k: g[k] * (base[k].pow(2).mean().sqrt() / (g[k].pow(2).mean().sqrt() + 1e-20)) for k in par}
    ana = sum(float((g[k].double() * d[k].double()).sum()) for k in par)
    for eps in [float(e) for e in os.environ.get("EPS", "3e-4,1e-3,3e-3").split(",")]:
        vals = []
        for sgn in (+1, -1):
            for k in par: params[k].copy_(base[k] + sgn * eps * d[k])
            vals.append(loss()[0])
        fd = (vals[0] - vals[1]) / (2 * eps)
        print(f"bn_batch={bn} trial {trial} eps {eps:.0e}: analytic {ana:+.5e}  central difference {fd:+.5e}  rel err {abs(fd-ana)/max(abs(ana),1e-9):.2e}")
    for k in par: params[k].copy_(base[k])
