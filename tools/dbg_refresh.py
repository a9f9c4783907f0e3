"""Refreshed (device re-pack) vs host-packed model: where do encoder outputs differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_amd import native, synth
from model.model import get_network
net = get_network("train").cuda()
net.load_state_dict({k: v for k, v in synth.synth_state_dict().items() if k.startswith(("encoder.", "decoder."))}, strict=False)
net._train_state()
torch.manual_seed(5)
with torch.no_grad():
    for name, t in net.state_dict(keep_vars=True).items():
        if not t.is_floating_point() or name.endswith("pos_table"):
            continue
        if name.endswith("running_var"):
            t.mul_(1.0 + 0.2 * torch.rand_like(t))
        else:
            t.add_(0.02 * t.abs().mean() * torch.randn_like(t))
net.mark_weights_changed()
net.eval()
video = synth.synth_video(2, 29, tag="video-lrw2").cuda()
tensors = {k: v.detach() for k, v in net.state_dict().items() if k.startswith(("encoder.", "decoder."))}
ref = native.NativeModel(); ref.load(tensors, list(tensors.keys()))
nmr = net.native_model()
md = lambda a, b: float((a - b).abs().max())
for x3 in (1, 0):
    nmr.set_option("frontend_x3", x3); ref.set_option("frontend_x3", x3)
    print("frontend_x3", x3, "op_frontend maxdiff", md(nmr.op_frontend(video), ref.op_frontend(video)), "encoder_fwd maxdiff", md(nmr.encoder_fwd(video), ref.encoder_fwd(video)))
for opt in ("fuse_trunk", "gemm_x3"):
    nmr.set_option(opt, 0); ref.set_option(opt, 0)
    print(opt, "= 0: encoder_fwd maxdiff", md(nmr.encoder_fwd(video), ref.encoder_fwd(video)))
