"""Where does the bf16 training leg's gradient leave the fp32 one?  One step at config 3's shape (B=8, S=77) in four settings
(eval / train() BatchNorm x dropout off / on), per-tensor gradient norms fp32 vs bf16 for the tensors that differ most."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_amd import synth
from lip2speech_amd.training import draw_dropout
from oracle import l2s_oracle as orc
from model.model import get_network

Bc, T, Sc = int(os.environ.get("B", 8)), 29, int(os.environ.get("S", 77))
tf = 0.5
video = synth.synth_video(Bc, T, tag="cfg3").cuda(); emb = synth.synth_speaker_embedding(Bc, tag="cfg3").cuda()
gum = synth.synth_gumbel(Bc * 4, tag="cfg3").cuda(); mels = synth.synth_mels(Bc, Sc, tag="cfg3").cuda()
gate = torch.zeros(Bc, Sc).cuda(); gate[:, -1] = 1.0
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith(("encoder.", "decoder."))}
drop = draw_dropout(Bc, T, Sc, "cuda", generator=torch.Generator("cuda").manual_seed(11))


def step(train, use_drop, bf16, tfr):
    net = get_network("train").cuda()
    net.load_state_dict(sd, strict=False)
    net.train(train)
    net._train_state()
    net.native_model().set_option("train_bf16", bf16)
    torch.manual_seed(3)
    kw = {"dropout_masks": drop} if use_drop else ({"dropout_masks": {}} if train else {})
    out = net(video, None, None, mels, torch.full((Bc,), T), None, None, tfr, speaker_embedding=emb, gumbel_noise=gum, **kw)
    terms = orc.loss_terms(out, mels, gate)
    terms[-1].backward()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, [float(t) for t in terms]


for train, use_drop, tfr in ((False, False, 1), (True, False, 1), (True, True, 1), (True, True, 0.5), (False, False, 0.5)):
    g32, l32 = step(train, use_drop, 0, tfr)
    g16, l16 = step(train, use_drop, 1, tfr)
    n32 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g32.values())))
    n16 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g16.values())))
    print(f"== train={train} dropout={use_drop} tf={tfr}: loss {l32[-1]:.4f} / {l16[-1]:.4f}; grad norm fp32 {n32:.3f} bf16 {n16:.3f}")
    rows = []
    for k in g32:
        a, b = float(g32[k].norm()), float(g16[k].norm())
        rows.append((abs(a - b), k, a, b))
    rows.sort(reverse=True)
    for d, k, a, b in rows[:8]:
        print(f"   {k:60s} {a:12.4f} {b:12.4f}")

# sensitivity of the FP32 step itself: the same step with the frames perturbed by bf16-sized relative noise (2^-9) - if the encoder / K-path
# gradient swings as much as it does between the fp32 and the bf16 leg, the difference above is the path's own conditioning, not a kernel
print("== fp32 step, train() + dropout + tf 0.5, frames perturbed by 2^-9 relative noise (three draws)")
base, _ = step(True, True, 0, 0.5)
v0 = video.clone()
for seed in (1, 2, 3):
    g = torch.Generator("cuda").manual_seed(seed)
    video = v0 * (1 + 2.0 ** -9 * torch.randn(v0.shape, device="cuda", generator=g))
    gp, lp = step(True, True, 0, 0.5)
    def nrm(d, pre): return float(torch.sqrt(sum((v.double() ** 2).sum() for k, v in d.items() if k.startswith(pre))))
    print(f"   seed {seed}: loss {lp[-1]:.4f}; |g| encoder {nrm(base, 'encoder.'):.2f} -> {nrm(gp, 'encoder.'):.2f}; decoder.K {nrm(base, 'decoder.K.'):.2f} -> {nrm(gp, 'decoder.K.'):.2f}; "
          f"postnet {nrm(base, 'decoder.postnet.'):.2f} -> {nrm(gp, 'decoder.postnet.'):.2f}; decoder_rnn {nrm(base, 'decoder.decoder_rnn.'):.2f} -> {nrm(gp, 'decoder.decoder_rnn.'):.2f}")
