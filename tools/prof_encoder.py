#!/usr/bin/env python
"""Run the visual encoder alone (for rocprofv3 --kernel-trace)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in sd.items() if k.startswith("encoder.")}, [k for k in sd if k.startswith("encoder.")])
v = synth.synth_video(32, 29, tag="bench").cuda()
for _ in range(6):
    nm.encoder_fwd(v)
torch.cuda.synchronize()
