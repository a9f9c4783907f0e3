#!/usr/bin/env python
"""Grouped decode check: l2s_inference over G*32 rows with the register-blocked step kernels against (a) the same rows with the 1x1
blocks and (b) the G batches run one by one - all three must agree bit for bit; then timings per option."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
T, S = 29, 300
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
G = int(os.environ.get("G", 4))
parts = [(synth.synth_video(32, T, tag=f"g{g}").cuda(), synth.synth_speaker_embedding(32, tag=f"g{g}").cuda(), synth.synth_gumbel(128, tag=f"g{g}").cuda()) for g in range(G)]
big = tuple(torch.cat([p[i] for p in parts]) for i in range(3))
nm.set_option("skinny_rc", 11)
ref = [nm.inference(*p, S=S, want_attn=True) for p in parts]
ref = tuple(torch.cat([r[i] for r in ref]) for i in range(3))
one = nm.inference(*big, S=S, want_attn=True)
print("B=%d 1x1 blocks vs %d x B=32: mel equal %s lengths equal %s attn equal %s" % (32 * G, G, torch.equal(one[0], ref[0]), torch.equal(one[1], ref[1]), torch.equal(one[2], ref[2])))
for shape in (0, 21, 22, 42):
    for jb in (4, 2):
        nm.set_option("skinny_rc", shape); nm.set_option("skinny_rc_jb", jb)
        out = nm.inference(*big, S=S, want_attn=True)
        ok = all(torch.equal(out[i], ref[i]) for i in range(3))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): nm.inference(*big, S=S)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        print(f"skinny_rc={shape:2d} jb={jb}: bit-identical to one-by-one: {ok}  max|d| {float((out[0]-ref[0]).abs().max()):.2e}   {dt*1e3:.2f} ms/pass = {dt/G*1e3:.2f} ms per B=32 batch  {32*G*S/dt/1e3:.0f} k mel-frames/s", flush=True)
