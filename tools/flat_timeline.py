#!/usr/bin/env python
"""Phase timeline of the decode step's FIRST launch at 256 rows (skinny_flat_kernel: prenet1 o fc_out, Q, content Q, fc_out + stop as one flat grid of
per-group block shapes) from its stamped build (`l2s_op_flat_timeline`): thread 0 of every block stamps the 100 MHz wall clock.  A grouped pass of
a few steps runs; the last first-phase launch leaves its stamps.
-> profiles/rNN_flat_timeline.txt"""
import os, sys, torch, numpy as np
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.set_option("flat_half", 0)      # the stamped build is the eight-wave form's (one block per CU)
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
G = 8
batches = [(synth.synth_video(32, 29, tag=f"b{i}").cuda(), synth.synth_speaker_embedding(32, tag=f"b{i}").cuda(), synth.synth_gumbel(32 * 4, tag=f"b{i}").cuda()) for i in range(G)]
nm.inference_multi(batches, S=4); torch.cuda.synchronize()
L = native.lib()
NB = 256
ts = torch.zeros(NB * 8, dtype=torch.int64, device="cuda")
native.check(L.l2s_op_flat_timeline(ts.data_ptr()))
nm.inference_multi(batches, S=4); torch.cuda.synchronize()
native.check(L.l2s_op_flat_timeline(None))
t = ts.cpu().numpy().reshape(NB, 8).astype(np.float64) * 0.01
live = t[:, 0] > 0
t = t[live]; n = int(live.sum())
t -= t[:, 0].min()
names = ["entry", "params in SGPRs", "loads issued", "first operands landed", "MFMAs done", "after reduction barrier"]      # the element epilogue that follows is not stamped
print(f"{n} blocks")
print(f"{'stamp':26s} {'min':>7s} {'median':>7s} {'max':>7s}   (us since the first block entered)")
for i, nme in enumerate(names): print(f"{nme:26s} {t[:, i].min():7.2f} {np.median(t[:, i]):7.2f} {t[:, i].max():7.2f}")
t = t[:, :6]
d = np.diff(t, axis=1)
print("per-block phase durations (median us):", " | ".join(f"{names[i+1]}: {np.median(d[:, i]):.2f}" for i in range(5)))
for lo in range(0, n, 32):
    seg = slice(lo, min(n, lo + 32))
    print(f"  blocks {lo:3d}-{min(n, lo + 32) - 1:3d} (longest groups first): entry {np.median(t[seg, 0]):5.2f}  loads issued {np.median(d[seg, 1]):5.2f}  landed {np.median(d[seg, 2]):5.2f}  wave 0's MFMAs {np.median(d[seg, 3]):5.2f}  wait for the other waves {np.median(d[seg, 4]):5.2f}  -> {np.median(t[seg, 5]):5.2f}")
