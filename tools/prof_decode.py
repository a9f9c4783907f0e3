#!/usr/bin/env python
"""Run only the decode loop (prologue once, then 3 x 300 steps at B=32) - for rocprofv3 counter passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B, T, S = 32, 29, 300
v = synth.synth_video(B, T, tag="bench").cuda()
emb = synth.synth_speaker_embedding(B, tag="bench").cuda()
gum = synth.synth_gumbel(B * 4, tag="bench").cuda()
feat = nm.encoder_fwd(v)
state, _ = nm.decoder_prologue(native.build_visual(feat, emb), emb, gum)
for _ in range(int(os.environ.get("REPS", "3"))):
    nm.decode_steps(state, B, T, S, want_attn=False)
torch.cuda.synchronize()
