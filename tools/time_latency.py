#!/usr/bin/env python
"""Latency of ONE l2s_inference call (T=29, S=300) for B = 1, 2 clips alone on the GPU: the launch-per-phase decode loop against the persistent
weight-stationary loop (option "persist_decode", pdecode.hip), whole call and decode loop alone (l2s_decode_steps on a prepared state), with the
deviation between the two paths.  ROWS env: comma list of B (default 1,2); REPS env: calls per median (default 7).
-> profiles/r04_latency_path.txt"""
import os, sys, time, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
rows = [int(x) for x in os.environ.get("ROWS", "1,2").split(",")]
reps = int(os.environ.get("REPS", 7)); S = int(os.environ.get("S", 300)); T = 29
sd = synth.synth_state_dict()
def model(persist):
    nm = native.NativeModel(); nm.set_option("persist_decode", persist)
    nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys())); return nm
base, pers = model(0), model(8)
def med(f):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
print(f"one call alone on the GPU, T={T}, S={S}; ms, median of {reps}")
print(f"{'B':>3s} | {'inference launch':>16s} {'persistent':>10s} {'ratio':>6s} | {'decode launch':>13s} {'persistent':>10s} {'us/step':>8s} {'us/step':>8s} | max|d mel_post| max|d attn|")
for B in rows:
    v = synth.synth_video(B, T, tag=f"lat{B}").cuda(); e = synth.synth_speaker_embedding(B, tag=f"lat{B}").cuda(); g = synth.synth_gumbel(B * native.min_T(T), tag=f"lat{B}").cuda()
    outs = {}
    res = []
    for name, nm in (("launch", base), ("persist", pers)):
        for _ in range(2): o = nm.inference(v, e, g, S=S, want_attn=True)
        outs[name] = [x.clone() for x in o]
        t_inf = med(lambda: nm.inference(v, e, g, S=S))
        feat = nm.encoder_fwd(v); vis = native.build_visual(feat, e); state, _ = nm.decoder_prologue(vis, e, g)
        for _ in range(2): nm.decode_steps(state, B, T, S)
        t_dec = med(lambda: nm.decode_steps(state, B, T, S))
        res.append((t_inf, t_dec))
    d_mel = (outs["launch"][0] - outs["persist"][0]).abs().max().item(); d_at = (outs["launch"][2] - outs["persist"][2]).abs().max().item()
    print(f"{B:3d} | {res[0][0]:16.3f} {res[1][0]:10.3f} {res[0][0] / res[1][0]:6.2f} | {res[0][1]:13.3f} {res[1][1]:10.3f} {res[0][1] * 1e3 / S:8.2f} {res[1][1] * 1e3 / S:8.2f} | {d_mel:.2e} {d_at:.2e}")
