"""A few launches of the front-end conv kernel (split-bf16 path, or f32 with X3=0) at B=128, for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.set_option("frontend_x3", int(os.environ.get("X3", 1))); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
v = synth.synth_video(32, 29, tag="bench").cuda().repeat(4, 1, 1, 1, 1)
for _ in range(4): nm.op_frontend(v)
torch.cuda.synchronize()
