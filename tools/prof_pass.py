"""Per-kernel-name GPU time of one inference pass (B=32, T=29, S=300), HIP-event brackets (inflates us-scale kernels by ~1.8 us each)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
B, T, S = 32, 29, 300
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
video = synth.synth_video(B, T, tag="bench").cuda(); emb = synth.synth_speaker_embedding(B, tag="bench").cuda(); gum = synth.synth_gumbel(B * 4, tag="bench").cuda()
for _ in range(3): nm.inference(video, emb, gum, S=S)
native.profile_enable(True); native.profile_reset()
n = 3
for _ in range(n): nm.inference(video, emb, gum, S=S)
torch.cuda.synchronize()
prof = sorted(native.profile_read(), key=lambda r: -r[2])
tot = sum(r[2] for r in prof) / n
print(f"sum of bracketed kernel times per pass: {tot:.3f} ms")
for name, cnt, ms in prof:
    if ms / n > 0.01: print(f"  {name:36s} {cnt//n:5d} launches {ms/n:8.3f} ms/pass  avg {ms/cnt*1e3:8.1f} us")
