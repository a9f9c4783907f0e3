#!/usr/bin/env python
"""Random (B, T, HW, S) shapes: HIP inference against the CPU oracle (mel, lengths, attention argmax where the margin is clear)."""
import os, sys, random, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
from oracle import l2s_oracle as orc
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
rng = random.Random(int(os.environ.get("SEED", 7)))
worst = 0.0
cases = [(1, 7, 96, 3), (3, 75, 88, 5), (17, 8, 96, 4), (33, 29, 96, 6), (2, 13, 88, 300)] + \
        [(rng.randint(1, 24), rng.randint(7, 60), rng.choice((88, 96)), rng.randint(1, 40)) for _ in range(int(os.environ.get("N", 8)))]
torch.set_num_threads(32)
for B, T, HW, S in cases:
    tag = f"fz{B}_{T}_{HW}_{S}"
    v = synth.synth_video(B, T, HW, HW, tag=tag); e = synth.synth_speaker_embedding(B, tag=tag); g = synth.synth_gumbel(B * native.min_T(T), tag=tag)
    mel, ln, at = nm.inference(v.cuda(), e.cuda(), g.cuda(), S=S, want_attn=True)
    with torch.no_grad():
        omel, oln, oat = orc.inference(sd, v, e, g, S=S)
    d = float((mel.cpu() - omel).abs().max())
    srt, idx = torch.sort(oat, dim=-1, descending=True)
    sure = (srt[..., 0] - srt[..., 1]) > 1e-4
    amax = at.cpu().argmax(-1)
    ok = bool(torch.equal(amax[sure], idx[..., 0][sure])) and bool(torch.equal(ln.cpu(), oln))
    worst = max(worst, d)
    print(f"B={B:3d} T={T:3d} HW={HW} S={S:3d}: mel max|d| {d:.2e}  lengths/argmax {'ok' if ok else 'MISMATCH'}", flush=True)
    assert d < 1e-3 and ok
print(f"worst mel deviation {worst:.2e}")
