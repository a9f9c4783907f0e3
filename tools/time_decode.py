#!/usr/bin/env python
"""Wall-clock of the decode loop alone (B=32, T=29, S=300), median of several runs."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B, T, S = 32, 29, 300
for kv in filter(None, os.environ.get("L2S_OPTS", "").split(",")):
    k, v = kv.split("="); nm.set_option(k, int(v))
v = synth.synth_video(B, T, tag="bench").cuda()
emb = synth.synth_speaker_embedding(B, tag="bench").cuda()
gum = synth.synth_gumbel(B * 4, tag="bench").cuda()
feat = nm.encoder_fwd(v)
state, _ = nm.decoder_prologue(native.build_visual(feat, emb), emb, gum)
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nm.decode_steps(state, B, T, S, want_attn=False)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"decode loop: median {ts[len(ts)//2]*1e3:.3f} ms  min {ts[0]*1e3:.3f} ms  ({ts[len(ts)//2]/S*1e6:.2f} us/step) opts {os.environ.get('L2S_OPTS')} lib {os.path.basename(native.LIB_PATH)}")
