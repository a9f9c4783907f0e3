#!/usr/bin/env python
"""Phase timeline of the decoder LSTM-cell kernel: the stamped build of skinny_kernel inside a chain of dependent launches."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B = 32
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
