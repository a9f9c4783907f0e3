"""SHA-256 of the visual encoder's features at B = 32 (88x88 and 96x96 crops) and B = 256 (the grouped shapes: five frames per 3x3 block) plus the
per-kernel-name times of the B = 256 pass: a pure data-movement change of the fused ShuffleNet units must not change the hashes.
L2S_LIB=<other build> python tools/hash_encoder.py for the A/B.
-> profiles/rNN_encoder_hash.txt"""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
print(os.path.basename(native.LIB_PATH))
for B, hw in ((32, 96), (32, 88), (256, 96)):
    v = synth.synth_video(32, 29, tag="enc").cuda()
    if hw == 88: v = v[..., 4:92, 4:92].contiguous()
    v = v.repeat(B // 32, 1, 1, 1, 1)
    if B > 32: v = v + 0.01 * torch.arange(B, device="cuda").view(B, 1, 1, 1, 1) / B          # distinct clips
    f = nm.encoder_fwd(v)
    torch.cuda.synchronize()
    print(f"  B={B} {hw}x{hw}: sha256 {hashlib.sha256(f.cpu().numpy().tobytes()).hexdigest()[:16]}  |feat| {f.abs().mean().item():.6f}")
native.profile_enable(True); native.profile_reset(); nm.encoder_fwd(v); torch.cuda.synchronize()
for n, l, t in sorted(native.profile_read(), key=lambda r: -r[2])[:9]: print(f"    {n:36s} {l:3d} x {1e3 * t / l:8.1f} us = {t:7.3f} ms")
