"""fp32 and bf16-operand training loss curves of the same 40 train()-mode steps (identical masks and sampling draws), and the fp32 curve's own
sensitivity: the same fp32 run from parameters perturbed by one part in 1e6."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import synth, callers
from model.model import get_network
T, Bb, Sb = 29, 4, 40
video = synth.synth_video(Bb, T, tag="bf16"); emb = synth.synth_speaker_embedding(Bb, tag="bf16")
mels = synth.synth_mels(Bb, Sb, tag="bf16"); gate = torch.zeros(Bb, Sb); gate[:, -1] = 1.0
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith(("encoder.", "decoder."))}
audio = torch.zeros(Bb, 256 * (Sb - 1))
batch = ((video, torch.full((Bb,), T)), (audio, torch.full((Bb,), audio.shape[1])), (mels, torch.full((Bb,), Sb), gate), None)
class Spk:
    def inference(self, a): return emb.to(a.device)
def run(bf16, eps=0.0):
    net = get_network("train").cuda()
    sdx = {k: (v * (1 + eps) if v.is_floating_point() and k.endswith("weight") else v) for k, v in sd.items()}
    net.load_state_dict(sdx, strict=False)
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    return np.array([r["loss"] for r in callers.train_iterations(net, [batch], 40, speaker_encoder=Spk(), tf_ratio=0.5, bf16=bf16)])
c32, c16, c32p = run(False), run(True), run(False, 1e-6)
np.set_printoptions(precision=1, linewidth=200, suppress=True)
print("fp32      ", c32); print("bf16      ", c16); print("fp32 (1+1e-6) w", c32p)
d = lambda a, b: np.abs(a - b) / b
print(f"bf16 vs fp32: max {d(c16, c32).max():.3f} mean {d(c16, c32).mean():.3f}; perturbed fp32 vs fp32: max {d(c32p, c32).max():.3f} mean {d(c32p, c32).mean():.3f}")
print(f"mean loss over the 40 steps: fp32 {c32.mean():.1f} bf16 {c16.mean():.1f} perturbed {c32p.mean():.1f}; last five: {c32[-5:].mean():.1f} {c16[-5:].mean():.1f} {c32p[-5:].mean():.1f}")
