#!/usr/bin/env python
"""Fused ShuffleNet units, split-bf16 pointwise convs (option trunk_x3 = 1) against the f32-MFMA units (0): encoder features against each other and
against the reference golden, and the encoder's per-kernel-name times at B clips (B env, default 256; HIP-event brackets of l2s_prof).
-> profiles/rNN_trunk_x3.txt"""
import os, sys, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from lip2speech_amd import native, synth
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith("encoder.")}
B = int(os.environ.get("B", 256))
v = synth.synth_video(32, 29, tag="bench").cuda().repeat(max(1, B // 32), 1, 1, 1, 1)[:B]
feats = {}
for x3 in ((1, 0) if os.environ.get("REVERSE") else (0, 1)):
    nm = native.NativeModel(); nm.load({k: v_.cuda() for k, v_ in sd.items()}, list(sd.keys()))
    nm.set_option("trunk_x3", x3)
    for _ in range(3): f = nm.encoder_fwd(v)
    feats[x3] = f.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): nm.encoder_fwd(v)
    e1.record(); torch.cuda.synchronize()
    print(f"trunk_x3={x3}: encoder {e0.elapsed_time(e1) / 5:.3f} ms per {B} clips")
    native.profile_enable(True); native.profile_reset()
    nm.encoder_fwd(v); torch.cuda.synchronize()
    tot = 0.0
    for name, n, ms in native.profile_read():
        if "shuffle" in name: print(f"    {name:36s} {ms * 1e3 / max(n, 1):8.1f} us x {n}"); tot += ms
    print(f"    all 16 units: {tot:.3f} ms")
    native.profile_enable(False)
d = (feats[0] - feats[1]).abs()
print(f"features x3 vs f32 units: max |d| {d.max().item():.3e}, mean {d.mean().item():.3e} (unit-norm rows of 768)")
try:
    import parity_common as pc
    g, video, _ = pc.lrw2_inputs()
    for x3 in (0, 1):
        nm = native.NativeModel(); nm.load({k: v_.cuda() for k, v_ in sd.items()}, list(sd.keys())); nm.set_option("trunk_x3", x3)
        print(f"trunk_x3={x3}: max |feat - reference golden| {(nm.encoder_fwd(video.cuda()).cpu() - torch.as_tensor(g['feat'])).abs().max().item():.3e}")
except Exception as e:      # noqa: BLE001
    print("golden check skipped:", e)
