"""A few encoder passes (B=32, T=29) for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = {k: v for k, v in synth.synth_state_dict().items() if k.startswith("encoder.")}
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
v = synth.synth_video(32, 29, tag="bench").cuda()
for _ in range(3): nm.encoder_fwd(v)
torch.cuda.synchronize()
