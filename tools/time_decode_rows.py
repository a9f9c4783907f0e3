"""Decode loop only (prologue once) at ROWS clips per launch, per option setting: us per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict(); tens = {k: v.cuda() for k, v in sd.items()}
T, S = 29, 300
for rows in [int(r) for r in os.environ.get("ROWS", "128,256").split(",")]:
    G = rows // 32
    v = synth.synth_video(32, T, tag="bench").cuda().repeat(G, 1, 1, 1, 1); emb = synth.synth_speaker_embedding(32, tag="bench").cuda().repeat(G, 1)
    gum = synth.synth_gumbel(128, tag="bench").cuda().repeat(G, 1)
    for opts in ({}, {"skinny_rc_multi": 21}, {"skinny_rc_multi": 22}, {"skinny_rc_multi": 42}, {"skinny_rc_multi": 11}):
        nm = native.NativeModel()
        for k, val in opts.items(): nm.set_option(k, val)
        nm.load(tens, list(sd.keys()))
        state, _ = nm.decoder_prologue(native.build_visual(nm.encoder_fwd(v), emb), emb, gum)
        nm.decode_steps(state, rows, T, S, want_attn=False); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); nm.decode_steps(state, rows, T, S, want_attn=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"rows={rows} {opts}: {min(ts)/S*1e6:.2f} us/step", flush=True)
        del nm
