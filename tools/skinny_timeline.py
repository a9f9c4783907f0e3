#!/usr/bin/env python
"""Phase timeline of the decoder LSTM-cell kernel: the stamped build of skinny_kernel inside a chain of dependent launches.
-> profiles/rNN_lstm_timeline_x3.txt (X3=2), rNN_lstm_timeline_4wave.txt (X3=1)"""
import os, sys, torch, numpy as np
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.set_option("lstm_x3", int(os.environ.get("X3", "2")))      # 2 (default): split-bf16 LSTM blocks on eight waves, 1: on four, 0: the f32 MFMA form
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B = int(os.environ.get("ROWS", "32"))      # ROWS=256: the straight-line 4x2 form (skinny_block_rcs), one block per CU
L = native.lib()
ts = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
print("production chain:", nm.lstm_cell_chain_us(B, 300), "us/launch")
native.check(L.l2s_op_skinny_timeline(ts.data_ptr()))
us = nm.lstm_cell_chain_us(B, 300)          # the last launch (layer 1, K = 1024) leaves its stamps
native.check(L.l2s_op_skinny_timeline(None))
print("stamped chain:", us, "us/launch")
t = ts.cpu().numpy().reshape(256, 8).astype(np.float64) * 0.01     # us
t0 = t[:, 0].min()
t -= t0
names = ["entry", "params in SGPRs", "loads issued", "first operands landed", "MFMAs done", "after reduction barrier", "after gate barrier", "stores drained"]
print(f"{'stamp':26s} {'min':>7s} {'median':>7s} {'max':>7s}   (us since the first block entered)")
for i, n in enumerate(names):
    print(f"{n:26s} {t[:, i].min():7.2f} {np.median(t[:, i]):7.2f} {t[:, i].max():7.2f}")
d = np.diff(t, axis=1)
print("per-block phase durations (median us):", " | ".join(f"{names[i+1]}: {np.median(d[:, i]):.2f}" for i in range(7)))
print(f"kernel span: {t[:, 7].max():.2f} us; block lifetime median {np.median(t[:, 7] - t[:, 0]):.2f} us")
if B >= 256:
    # 4x2 blocks: block index = y * 64 + x (x = column pair 0..63 -> XCD x % 8, y = row group 0..3); which blocks are late?
    x, y = np.arange(256) % 64, np.arange(256) // 64
    done = t[:, 4]
    print("MFMAs-done by XCD (median / max us):", " | ".join(f"{c}: {np.median(done[x % 8 == c]):.1f}/{done[x % 8 == c].max():.1f}" for c in range(8)))
    print("MFMAs-done by row group (median / max):", " | ".join(f"{r}: {np.median(done[y == r]):.1f}/{done[y == r].max():.1f}" for r in range(4)))
    order = np.argsort(-t[:, 7])[:12]
    print("latest blocks (x, y, xcd): entry / loads issued / first landed / MFMAs done / end")
    for b in order:
        print(f"   ({x[b]:2d},{y[b]},{x[b] % 8})  {t[b,0]:5.2f} {t[b,2]:5.2f} {t[b,3]:5.2f} {t[b,4]:6.2f} {t[b,7]:6.2f}")
    print("wave-0 compute time (first landed -> MFMAs done) percentiles 10/50/90:", np.percentile(t[:, 4] - t[:, 3], [10, 50, 90]).round(2),
          "; barrier wait (MFMAs done -> after reduction barrier) 10/50/90:", np.percentile(t[:, 5] - t[:, 4], [10, 50, 90]).round(2))
