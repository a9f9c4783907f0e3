#!/usr/bin/env python
"""Throughput of the other BASELINE.json shapes (parity-test cases, not bench lines): GRID-like B=16/T=75 and AVSpeech-like B=32/T=50,
`Lip2Speech.inference` (S=300) and the evaluate path `forward(tf_ratio=1)` with S = the clip's mel length; four batches in flight."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
from lip2speech_amd.parallel import InflightPool
sd = synth.synth_state_dict()
pool = InflightPool({k: v.cuda() for k, v in sd.items()}, list(sd.keys()), n_inflight=4)
N = 96
for name, B, T, Sf in (("LRW  B=32 T=29", 32, 29, 77), ("GRID B=16 T=75", 16, 75, 188), ("AVSp B=32 T=50", 32, 50, 126)):
    batch = (synth.synth_video(B, T, tag=name).cuda(), synth.synth_speaker_embedding(B, tag=name).cuda(), synth.synth_gumbel(B * native.min_T(T), tag=name).cuda())
    for label, S, fn in (("inference S=300", 300, None), (f"forward   S={Sf}", Sf, (lambda S_: (lambda m, b: m.forward_eval(b[0], b[1], b[2], S_)))(Sf))):
        pool.map([batch] * 8, S=S, fn=fn)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pool.map([batch] * N, S=S, fn=fn)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}  {label}: {dt / N * 1e3:6.2f} ms/batch  {B * S * N / dt / 1e3:8.1f} k mel-frames/s  {B * T * N / dt / 1e3:7.1f} k video frames/s", flush=True)
