"""Two inference passes (B=32, T=29, S=300) for rocprofv3 --pmc runs over the whole path."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
v = synth.synth_video(32, 29, tag="bench").cuda(); e = synth.synth_speaker_embedding(32, tag="bench").cuda(); g = synth.synth_gumbel(128, tag="bench").cuda()
for _ in range(2): nm.inference(v, e, g, S=300)
torch.cuda.synchronize()
