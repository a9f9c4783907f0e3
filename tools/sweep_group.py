#!/usr/bin/env python
"""Throughput of parallel.InflightPool over (batches per chain G) x (chains in flight): B=32, T=29, S=300, distinct batches per slot."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import synth
from lip2speech_amd.parallel import InflightPool
B, T, S = 32, 29, 300
sd = synth.synth_state_dict()
tens = {k: v.cuda() for k, v in sd.items()}
NB = 16
batches = [(synth.synth_video(B, T, tag=f"sw{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"sw{i}").cuda(), synth.synth_gumbel(B * 4, tag=f"sw{i}").cuda()) for i in range(NB)]
combos = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("COMBOS", "1x1,1x4,2x2,4x1,4x2,4x3,8x1,8x2").split(",")]
for G, NI in combos:
    pool = InflightPool(tens, list(sd.keys()), n_inflight=NI, group=G)
    n = max(2 * G * NI, 48) // (G * NI) * (G * NI)
    work = [batches[i % NB] for i in range(n)]
    pool.map(work[:G * NI], S=S)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pool.map(work, S=S)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"G={G} chains={NI}: {best/n*1e3:6.2f} ms per B=32 batch  {B*S*n/best/1e3:8.1f} k mel-frames/s  ({n} batches)", flush=True)
    del pool
