#!/usr/bin/env python
"""What does advancing G independent B=32 batches per launch chain buy?  `l2s_inference` over B = 32*G rows, one chain at a time and
NT chains in flight, with the HIP-event per-kernel breakdown of one pass (tools: G env = comma list, NT env = chains in flight).
-> profiles/rNN_group8_breakdown.txt (G=8 NT=2)"""
import os, sys, time, threading
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
T, S = 29, 300
for kv in filter(None, os.environ.get("L2S_OPTS", "").split(",")):
    k, v = kv.split("="); native.set_option(k, int(v))
sd = synth.synth_state_dict()
tens = {k: v.cuda() for k, v in sd.items()}
NT = int(os.environ.get("NT", 2))
models = []
for i in range(NT):
    nm = native.NativeModel(); nm.load(tens, list(sd.keys())); models.append(nm)
streams = [torch.cuda.Stream() for _ in range(NT)]
for G in [int(g) for g in os.environ.get("G", "1,2,4").split(",")]:
    B = 32 * G
    inp = (synth.synth_video(32, T, tag="bench").cuda().repeat(G, 1, 1, 1, 1), synth.synth_speaker_embedding(32, tag="bench").cuda().repeat(G, 1),
           synth.synth_gumbel(32 * 4, tag="bench").cuda().repeat(G, 1))
    def run(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n): models[i].inference(*inp, S=S)
    for i in range(NT): run(i, 2)
    torch.cuda.synchronize()
    n = max(4, 24 // G)
    t0 = time.perf_counter(); run(0, n); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print(f"G={G} (B={B}) one chain: {t1/n*1e3:.2f} ms/pass = {t1/n/G*1e3:.2f} ms per B=32 batch  {B*S*n/t1/1e3:.1f} k mel-frames/s", flush=True)
    th = [threading.Thread(target=run, args=(i, n)) for i in range(NT)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print(f"G={G} {NT} chains in flight: {t2/n/NT/G*1e3:.2f} ms per B=32 batch  {NT*B*S*n/t2/1e3:.1f} k mel-frames/s", flush=True)
    native.profile_enable(True); native.profile_reset()
    with torch.cuda.stream(streams[0]):
        models[0].inference(*inp, S=S)
    torch.cuda.synchronize()
    prof = sorted(native.profile_read(), key=lambda r: -r[2])
    native.profile_enable(False)
    tot = sum(r[2] for r in prof)
    print(f"  event-bracketed kernel time {tot:.2f} ms:")
    for name, launches, ms in prof[:int(os.environ.get("TOP", 14))]:
        print(f"    {name:40s} {launches:5d} x {ms/launches*1e3:8.1f} us = {ms:7.3f} ms")
