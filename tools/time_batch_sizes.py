import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
for B in (32, 64, 96):
    v = synth.synth_video(B, 29, tag="b").cuda(); e = synth.synth_speaker_embedding(B, tag="b").cuda(); g = synth.synth_gumbel(B * 4, tag="b").cuda()
    for _ in range(2): nm.inference(v, e, g, S=300)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): nm.inference(v, e, g, S=300)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"B={B}: {dt*1e3:.2f} ms/pass  {B*300/dt/1e3:.1f} k mel-frames/s")
