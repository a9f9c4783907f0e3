#!/usr/bin/env python
"""Is the ~3 us/launch chain cost the host's submission rate or the GPU's dependent-dispatch latency?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native
L = native.lib()
buf = torch.randn(64 * 1024 * 1024, device="cuda")
out = torch.zeros(4096, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for kind, blocks, npb, label in ((0, 256, 0, "empty 256 blocks"), (1, 256, 6, "touch 256 blocks 6x8KiB"), (1, 128, 48, "touch 128 blocks 48x8KiB (48 MB)")):
    for n in (1200, 4800):
        best = None
        for _ in range(4):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
            native.check(L.l2s_op_launch_chain(kind, n, blocks, npb, buf.data_ptr(), out.data_ptr(), s))
            t1 = time.perf_counter()
            e1.record()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rec = ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, e0.elapsed_time(e1) * 1e3 / n)
            best = rec if best is None or rec[1] < best[1] else best
        print(f"{label:36s} n={n}: host enqueue {best[0]:.2f} us/launch, wall {best[1]:.2f}, GPU (events) {best[2]:.2f}")
print("--- kernels of known duration (every block spins on the 100 MHz clock): wall per launch minus the spin = GPU-side gap")
for blocks in (256, 128):
    for ticks in (400, 600, 1000):
        n = 1200
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            native.check(L.l2s_op_launch_chain(2, n, blocks, ticks, buf.data_ptr(), out.data_ptr(), s))
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
        print(f"spin {ticks/100:.0f} us, {blocks} blocks: {best:.2f} us/launch -> gap {best - ticks/100:.2f} us")
