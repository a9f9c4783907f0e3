#!/usr/bin/env python
"""The bf16 leg of inference (model option "infer_bf16") against the fp32 path on the same inputs: encoder features, the decoder
prologue's keys / values, and how the mel frames drift along the 300 steps (the recurrent loop is fp32 in both: it only sees bf16-rounded
inputs).  Then the rate of a grouped pass (G env, default 8)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
T, S = 29, 300
sd = synth.synth_state_dict(); tens = {k: v.cuda() for k, v in sd.items()}
def model(**opt):
    nm = native.NativeModel()
    for k, v in opt.items(): nm.set_option(k, v)
    nm.load(tens, list(sd.keys())); return nm
f32, b16 = model(), model(infer_bf16=1)
B = 32
video = synth.synth_video(B, T, tag="bench").cuda(); emb = synth.synth_speaker_embedding(B, tag="bench").cuda(); gum = synth.synth_gumbel(B * 4, tag="bench").cuda()
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
rms = lambda a, b: ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
fa, fb = f32.encoder_fwd(video), b16.encoder_fwd(video)
print(f"encoder features: max rel {rel(fb, fa):.2e}  rms rel {rms(fb, fa):.2e}")
va, vb = native.build_visual(fa, emb), native.build_visual(fa, emb)       # the same fp32 features into both prologues
sa, da = f32.decoder_prologue(va, emb, gum); sb, db = b16.decoder_prologue(vb, emb, gum)
print(f"prologue state blob (k, v, content key / value, h0, encoder_cell): max rel {rel(sb, sa):.2e}  rms rel {rms(sb, sa):.2e};  content_dis max abs {(da - db).abs().max().item():.2e}")
ma, la = f32.inference(video, emb, gum, S=S)[:2]; mb, lb = b16.inference(video, emb, gum, S=S)[:2]
for lo, hi in ((0, 10), (10, 40), (40, 77), (77, 150), (150, 300)):
    d = (ma[:, :, lo:hi] - mb[:, :, lo:hi]).abs()
    print(f"mel frames {lo:3d}-{hi:3d}: max |d| {d.max().item():.3e}  mean |d| {d.mean().item():.3e}   (fp32 mel: mean |x| {ma[:, :, lo:hi].abs().mean().item():.3f})")
print(f"mel statistics fp32 mean {ma.mean().item():.4f} std {ma.std().item():.4f} | bf16 leg mean {mb.mean().item():.4f} std {mb.std().item():.4f}; lengths equal: {torch.equal(la, lb)}")
G = int(os.environ.get("G", 8))
batches = [(video, emb, gum)] * G
for name, nm in (("fp32", f32), ("bf16 leg", b16)):
    for _ in range(2): nm.inference_multi(batches, S=S)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 4
    for _ in range(n): nm.inference_multi(batches, S=S)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: G={G} one chain {dt*1e3:.2f} ms per pass = {G*B*S/dt/1e3:.0f} k mel-frames/s")
native.profile_enable(True); native.profile_reset(); b16.inference_multi(batches, S=S); torch.cuda.synchronize()
prof = sorted(native.profile_read(), key=lambda r: -r[2]); native.profile_enable(False)
for name, launches, ms in prof[:12]: print(f"    {name:40s} {launches:5d} x {ms/launches*1e3:8.1f} us = {ms:7.3f} ms")
