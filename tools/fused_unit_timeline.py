#!/usr/bin/env python
"""Phase timeline of the fused stride-1 ShuffleNet units (B env, default 32, T=29): the stamped build of one stage at a time; the last launch of
that stage leaves its stamps.
-> profiles/rNN_fused_unit_timeline_256clips.txt (B=256)"""
import os, sys, torch, numpy as np
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith("encoder.")}
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B = int(os.environ.get("B", 32))
v = synth.synth_video(32, 29, tag="bench").cuda().repeat(B // 32, 1, 1, 1, 1)
for _ in range(3): nm.encoder_fwd(v)
L = native.lib()
names = ["entry", "input in LDS", "after barrier", "pw1 done", "dw taps done", "dw written", "pw2 done", "stores drained"]
for h, nblk in ((12, 29 * B), (6, 29 * B // 2), (3, 29 * B // 2)):
    ts = torch.zeros(nblk * 10, dtype=torch.int64, device="cuda")
    native.check(L.l2s_op_fused_unit_timeline(ts.data_ptr(), h))
    nm.encoder_fwd(v); torch.cuda.synchronize()
    native.check(L.l2s_op_fused_unit_timeline(None, 0))
    raw = ts.cpu().numpy().reshape(nblk, 10)
    t = raw[:, :8].astype(np.float64) * 0.01
    t -= t[:, 0].min()
    print(f"--- {h}x{h} stage: {nblk} blocks")
    print(f"{'stamp':18s} {'min':>7s} {'median':>7s} {'max':>7s}   (us since the first block entered)")
    for i, n in enumerate(names): print(f"{n:18s} {t[:, i].min():7.2f} {np.median(t[:, i]):7.2f} {t[:, i].max():7.2f}")
    d = np.diff(t, axis=1)
    print("per-block phase durations (median us):", " | ".join(f"{names[i+1]}: {np.median(d[:, i]):.2f}" for i in range(7)))
    life = t[:, 7] - t[:, 0]
    print(f"block lifetime: median {np.median(life):.2f} us, min {life.min():.2f}, max {life.max():.2f}; kernel span {t[:, 7].max():.2f} us")
    cu = (raw[:, 8] >> 8) & 0xF; se = (raw[:, 8] >> 13) & 0x7; xcc = raw[:, 9] & 0xF
    u, cnt = np.unique(xcc * 1000 + se * 16 + cu, return_counts=True)
    print(f"distinct (XCC, SE, CU) ids: {len(u)}; blocks per id: min {cnt.min()} max {cnt.max()}")
    print("entry time percentiles (us):", " ".join(f"{q}%: {np.percentile(t[:, 0], q):.1f}" for q in (10, 25, 50, 75, 90, 100)))
