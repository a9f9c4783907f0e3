#!/usr/bin/env python
"""Phase timeline of the fused stride-2 ShuffleNet units (B env, default 256, T=29) from the stamped build (`l2s_op_fused_unit_timeline(ts, -H)`,
H = the unit's input size): thread 0 of every block stamps the 100 MHz wall clock after each phase.
-> profiles/rNN_s2_unit_timeline_256clips.txt"""
import os, sys, torch, numpy as np
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith("encoder.")}
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B = int(os.environ.get("B", 256))
v = synth.synth_video(32, 29, tag="bench").cuda().repeat(B // 32, 1, 1, 1, 1)
for _ in range(3): nm.encoder_fwd(v)
L = native.lib()
names = ["entry", "input + zero fill issued", "after barrier", "banch1 depthwise", "banch2 pw1 (input resolution)", "banch2 depthwise + barrier", "banch1 pw", "banch2 pw2", "stores drained"]
for h, strips in ((24, 6), (12, 3), (6, 1)):
    nblk = 29 * B * strips
    ts = torch.zeros(nblk * 10, dtype=torch.int64, device="cuda")
    native.check(L.l2s_op_fused_unit_timeline(ts.data_ptr(), -h))
    nm.encoder_fwd(v); torch.cuda.synchronize()
    native.check(L.l2s_op_fused_unit_timeline(None, 0))
    t = ts.cpu().numpy().reshape(nblk, 10)[:, :9].astype(np.float64) * 0.01
    t -= t[:, 0].min()
    d = np.diff(t, axis=1)
    print(f"--- stride-2 unit on the {h}x{h} map: {nblk} blocks; kernel span {t[:, 8].max():.1f} us; block lifetime median {np.median(t[:, 8] - t[:, 0]):.2f} us")
    print("    phase durations (median us): " + " | ".join(f"{n}: {np.median(d[:, i]):.2f}" for i, n in enumerate(names[1:])))
