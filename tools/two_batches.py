"""Two independent B=32 batches in flight on two streams (two host threads): does a second dependency chain fill the idle phases
(kernel boundaries, ramp, drain) of the first?  Serving-style concurrency; the headline bench stays one batch at a time."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
B, T, S = 32, 29, 300
sd = synth.synth_state_dict()
NT = int(os.environ.get("NT", 2))
for kv in filter(None, os.environ.get("L2S_OPTS", "").split(",")):
    k, v = kv.split("="); native.set_option(k, int(v))
models, inputs, streams = [], [], []
for i in range(NT):
    nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
    models.append(nm)
    inputs.append((synth.synth_video(B, T, tag=f"b{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"b{i}").cuda(), synth.synth_gumbel(B * 4, tag=f"b{i}").cuda()))
    streams.append(torch.cuda.Stream())
def run(i, n):
    with torch.cuda.stream(streams[i]):
        for _ in range(n): models[i].inference(*inputs[i], S=S)
for i in range(NT): run(i, 2)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter(); run(0, n); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print(f"one batch at a time: {t1/n*1e3:.2f} ms/pass  {B*S*n/t1/1e3:.1f} k mel-frames/s")
th = [threading.Thread(target=run, args=(i, n)) for i in range(NT)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize(); t2 = time.perf_counter() - t0
print(f"{NT} batches in flight ({NT} host threads, {NT} streams): {t2/n*1e3:.2f} ms per {NT} passes  {NT*B*S*n/t2/1e3:.1f} k mel-frames/s  ({NT*t1/t2:.2f}x)")
