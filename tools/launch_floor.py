#!/usr/bin/env python
"""What does one dependent launch cost on this GPU?  Chains of 1200 launches (= 300 decode steps x 4 phases)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native
L = native.lib()
buf = torch.randn(64 * 1024 * 1024, device="cuda")          # 256 MB source
out = torch.zeros(4096, device="cuda")
def chain(kind, blocks, npb, n=1200):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        native.check(L.l2s_op_launch_chain(kind, n, blocks, npb, buf.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / n * 1e6
print(f"empty kernel, 256 blocks x 512 thr : {chain(0, 256, 0):.2f} us/launch")
print(f"empty kernel, 32 blocks            : {chain(0, 32, 0):.2f} us/launch")
for npb in (1, 4, 6, 12):
    mb = 256 * npb * 8 / 1024
    print(f"touch kernel, 256 blocks, {npb:2d} x 8 KiB per block ({mb:5.1f} MB/launch): {chain(1, 256, npb):.2f} us/launch")
print(f"touch kernel, 128 blocks, 12 x 8 KiB: {chain(1, 128, 12):.2f} us/launch")
print(f"touch kernel, 512 blocks,  6 x 8 KiB: {chain(1, 512, 6):.2f} us/launch")

# two independent chains issued alternately on two streams: aggregate cost per launch PAIR
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def chain2(kind, blocks, npb, n=1200):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        native.check(L.l2s_op_launch_chain2(kind, n, blocks, npb, buf.data_ptr(), out.data_ptr(), sa.cuda_stream, sb.cuda_stream))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / n * 1e6
print("--- two streams (us per PAIR of launches, one on each stream)")
print(f"empty kernel, 128 blocks x2 streams: {chain2(0, 128, 0):.2f} us/pair   (single stream: {chain(0, 128, 0):.2f} us/launch)")
for blocks, npb in ((128, 6), (128, 12), (256, 3), (256, 6)):
    print(f"touch kernel, {blocks} blocks, {npb:2d} x 8 KiB x2 streams: {chain2(1, blocks, npb):.2f} us/pair   (single stream, same kernel: {chain(1, blocks, npb):.2f} us/launch; "
          f"single stream, double-size kernel: {chain(1, blocks * 2 if blocks < 256 else blocks, npb if blocks < 256 else npb * 2):.2f})")
