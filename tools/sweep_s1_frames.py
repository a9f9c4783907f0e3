"""Frames-per-block sweep of the fused stride-1 ShuffleNet units (B=32, T=29 -> 928 frames)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in sd.items() if k.startswith("encoder.")}, [k for k in sd if k.startswith("encoder.")])
v = synth.synth_video(32, 29, tag="bench").cuda()
ref = nm.encoder_fwd(v).clone()
for opt, name, cands in (("s1_frames_h12", "shuffle_unit_s1_fused_h12", (1,)), ("s1_frames_h6", "shuffle_unit_s1_fused_h6", (1, 2, 3, 4)),
                         ("s1_frames_h3", "shuffle_unit_s1_fused_h3", (1, 2, 3, 4, 5, 7, 8))):
    for F in cands:
        native.set_option(opt, F)
        try:
            out = nm.encoder_fwd(v)
        except Exception as e:
            print(opt, F, "unsupported:", str(e)[:60]); continue
        same = torch.equal(out, ref)
        native.profile_enable(True); native.profile_reset()
        for _ in range(5): nm.encoder_fwd(v)
        torch.cuda.synchronize()
        r = [x for x in native.profile_read() if x[0] == name][0]
        native.profile_enable(False)
        print(f"{opt} F={F}: {r[2] / r[1] * 1e3:7.1f} us/unit  ({r[1] // 5} units)  bit-identical to default: {same}")
    native.set_option(opt, 0)
