"""Per-kernel-name GPU time of one train()-mode training step (B=8, T=29, S=77).
-> profiles/rNN_train_kernels.txt"""
import os, sys, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
from lip2speech_amd.training import model_forward_backward, draw_dropout
B, T, S = int(os.environ.get("B", 8)), 29, 77
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
bound = {k: v.clone().cuda() for k, v in sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point()}
is_buf = lambda k: k.endswith(("running_mean", "running_var", "pos_table"))
grads = {k: torch.zeros_like(v) for k, v in bound.items() if not is_buf(k)}
nm.train_bind(bound, grads); nm.train_set_bn(True, 0.1)
video = synth.synth_video(B, T, tag="tt").cuda(); emb = synth.synth_speaker_embedding(B, tag="tt").cuda()
gum = synth.synth_gumbel(B * 4, tag="tt").cuda(); mels = synth.synth_mels(B, S, tag="tt").cuda()
gate = torch.zeros(B, S, device="cuda"); gate[:, -1] = 1
def step(): return model_forward_backward(nm, video, emb, gum, mels, gate, drop=draw_dropout(B, T, S, "cuda"))
for _ in range(2): step()
native.profile_enable(True); native.profile_reset(); step(); torch.cuda.synchronize()
prof = sorted(native.profile_read(), key=lambda r: -r[2]); tot = sum(r[2] for r in prof)
print(f"bracketed GPU time {tot:.2f} ms over {sum(r[1] for r in prof)} launches")
for name, cnt, ms in prof[:int(os.environ.get("TOP", 28))]: print(f"  {name:40s} {cnt:6d} launches {ms:8.3f} ms  {100*ms/tot:5.1f}%")
