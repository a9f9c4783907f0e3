"""Where does `evaluate.py`'s wall time go (evaluate.py:22-51)?  `callers.evaluate_net` over N loader batches of B=32 LRW-shaped clips (S = 77
targets): wall seconds waiting for the model (grouped HIP path, chains in flight), in the vocoder (`MelSpec2Audio`: InverseMelScale SGD
+ Griffin-Lim, 256 iterations each, torch ops on the device) and in ESTOI on the host (numpy) - VERDICT r2 item 9: time f4 before building it.
Writes the stage table to stdout (commit under profiles/)."""
import os
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lip2speech_amd import callers, synth
from model.model import get_network

N = int(os.environ.get("N", 8))
B, T, S = 32, 29, 77
ITERS = int(os.environ.get("ITERS", 256))
net = get_network("test")
net.load_state_dict(synth.synth_state_dict(), strict=True)
net = net.cuda()
gen = torch.Generator().manual_seed(0)
t = torch.arange(19456) / 16000.0
batches = []
for i in range(N):
    f0 = 120 + 10 * torch.arange(B).float()[:, None] + i
    audio = 0.2 * torch.sin(2 * np.pi * f0 * t[None]) + 0.02 * torch.randn(B, 19456, generator=gen)
    batches.append(((synth.synth_video(B, T, tag=f"ev{i % 4}"), torch.full((B,), T)), (audio, torch.full((B,), 19456)),
                    (synth.synth_mels(B, S, tag=f"ev{i}"), torch.full((B,), S), torch.zeros(B, S)), None))


class Spk:
    def inference(self, a):
        return synth.synth_speaker_embedding(B, tag="ev").to(a.device)


for group, inflight in ((8, 2), (1, 1)):
    callers.evaluate_net(net, batches[:2], speaker_encoder=Spk(), max_iters=4, group=group, n_inflight=inflight)     # warm-up
    tm = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    score = callers.evaluate_net(net, batches, speaker_encoder=Spk(), max_iters=ITERS, group=group, n_inflight=inflight, timings=tm)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"evaluate_net: {N} batches x B={B} (T={T}, S={S}), vocoder {ITERS}+{ITERS} iterations, group={group}, chains={inflight}: wall {wall:.2f} s, mean ESTOI {score:.4f}")
    for k in ("model_wait_s", "vocoder_s", "estoi_s"):
        print(f"    {k:14s} {tm[k]:8.3f} s  {100 * tm[k] / wall:5.1f} %   ({1e3 * tm[k] / tm['clips']:.2f} ms per clip)")
# the model alone, same batches (what the wait would be with no post-processing in the loop)
torch.cuda.synchronize()
t0 = time.perf_counter()
callers.evaluate_mels(net, batches, speaker_encoder=Spk())
torch.cuda.synchronize()
print(f"evaluate_mels alone (group 8, 2 chains): {time.perf_counter() - t0:.3f} s for {N} batches = {1e3 * (time.perf_counter() - t0) / N:.2f} ms per batch")
