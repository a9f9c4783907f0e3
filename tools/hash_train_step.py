"""SHA-256 over the outputs, all parameter gradients and the updated running statistics of ONE train()-mode forward + backward (B=8, T=29, S=77,
fixed masks): a change that is meant to be arithmetic-neutral (launch restructuring) must leave this hash unchanged.  L2S_LIB selects the build.
-> profiles/rNN_train_hash.txt"""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth, training
B, T, S = 8, 29, 77
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
bound = {k: v.clone().cuda() for k, v in sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point()}
is_buf = lambda k: k.endswith(("running_mean", "running_var", "pos_table"))
grads = {k: torch.zeros_like(v) for k, v in bound.items() if not is_buf(k)}
nm.train_bind(bound, grads); nm.train_set_bn(True, 0.1)
video = synth.synth_video(B, T, tag="tt").cuda(); emb = synth.synth_speaker_embedding(B, tag="tt").cuda()
gum = synth.synth_gumbel(B * 4, tag="tt").cuda(); mels = synth.synth_mels(B, S, tag="tt").cuda()
gate = torch.zeros(B, S, device="cuda"); gate[:, -1] = 1
torch.manual_seed(3); drop = training.draw_dropout(B, T, S, "cuda")
out = training.model_forward_backward(nm, video, emb, gum, mels, gate, drop=drop)
torch.cuda.synchronize()
h = hashlib.sha256()
for k in ("loss", "mel", "mel_post", "stop"): h.update(out[k].detach().cpu().numpy().tobytes())
for k in sorted(grads): h.update(grads[k].cpu().numpy().tobytes())
for k in sorted(bound):
    if k.endswith(("running_mean", "running_var")): h.update(bound[k].cpu().numpy().tobytes())
print(os.environ.get("L2S_LIB", "default"), h.hexdigest()[:16], out["loss"].cpu().tolist())
