#!/usr/bin/env python
"""Run only the decode loop (prologue once, then REPS x 300 steps over ROWS clips; ROWS = 32 x batches per launch chain, default 256) - for
rocprofv3 counter passes.
-> profiles/rNN_kernel_stats_decode256.md, rNN_pmc_decode.json (through tools/profile_r5.sh)"""
import os, sys, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
for kv in os.environ.get("L2S_OPT", "").split(","):      # e.g. L2S_OPT=skinny_rc_jb=2,skinny_flat=0
    if kv:
        nm.set_option(kv.split("=")[0], int(kv.split("=")[1]))
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B, T, S = int(os.environ.get("ROWS", "256")), 29, 300
G = B // 32
v = synth.synth_video(32, T, tag="bench").cuda().repeat(G, 1, 1, 1, 1)
emb = synth.synth_speaker_embedding(32, tag="bench").cuda().repeat(G, 1)
gum = synth.synth_gumbel(32 * 4, tag="bench").cuda().repeat(G, 1)
feat = nm.encoder_fwd(v)
state, _ = nm.decoder_prologue(native.build_visual(feat, emb), emb, gum)
for _ in range(int(os.environ.get("REPS", "3"))):
    nm.decode_steps(state, B, T, S, want_attn=False)
torch.cuda.synchronize()
print(f"decode loop: {B} rows, S={S}, done (run under rocprofv3: tools/profile_r5.sh)")
