#!/usr/bin/env python
"""Does the relative phase of two launch chains matter?  Two threads, each looping inference_multi over G=8 batches; thread 1 starts
`delay` ms after thread 0.  Also three chains."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
B, T, S, G = 32, 29, 300, int(os.environ.get("G", 8))
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
def mk(tag): return [(synth.synth_video(B, T, tag=f"{tag}{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"{tag}{i}").cuda(), synth.synth_gumbel(B * 4, tag=f"{tag}{i}").cuda()) for i in range(G)]
groups = [mk("a"), mk("b"), mk("c")]
streams = [torch.cuda.Stream() for _ in range(3)]
def run(i, n, delay):
    time.sleep(delay)
    with torch.cuda.stream(streams[i]):
        for _ in range(n): nm.inference_multi(groups[i], S=S)
for i in range(3): run(i, 1, 0)
torch.cuda.synchronize()
for nch, delays in ((1, [0]), (2, [0, 0]), (2, [0, 0.010]), (2, [0, 0.019]), (2, [0, 0.028]), (3, [0, 0, 0]), (3, [0, 0.013, 0.026])):
    n = 12
    th = [threading.Thread(target=run, args=(i, n, delays[i])) for i in range(nch)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"chains={nch} start delays {delays}: {dt/(n*nch*G)*1e3:.3f} ms per batch  {nch*n*G*B*S/dt/1e3:.0f} k mel-frames/s", flush=True)
