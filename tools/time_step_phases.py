"""Decode-step phase times at G batches per chain (rows = 32 G): HIP-event brackets per kernel name over one grouped pass (l2s_profile_*),
A/B over a run-time option.  Usage: python tools/time_step_phases.py [G] [option=value ...]   (each option is toggled against the default)
-> profiles/rNN_step_phases.txt"""
import os, sys
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_amd import native, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opts = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[2:]]      # "a=1,b=2" = one setting with two options
B, T, S = 32, 29, 300
sd = synth.synth_state_dict()
tensors = {k: v.cuda() for k, v in sd.items()}
batches = [(synth.synth_video(B, T, tag=f"b{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"b{i}").cuda(), synth.synth_gumbel(B * 4, tag=f"b{i}").cuda()) for i in range(G)]


def run(setting):
    nm = native.NativeModel()
    for k, v in setting.items():
        nm.set_option(k, v)
    nm.load(tensors, list(sd.keys()))
    for _ in range(2):
        out = nm.inference_multi(batches, S=S)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(3):
        out = nm.inference_multi(batches, S=S)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 3
    native.profile_enable(True); native.profile_reset()
    nm.inference_multi(batches, S=S)
    torch.cuda.synchronize()
    prof = {n: (l, t) for n, l, t in native.profile_read()}
    native.profile_enable(False)
    return ms, prof, out[0][0].clone()

base_ms, base_prof, base_out = run({})
print(f"G={G} ({32 * G} rows): default {base_ms:.3f} ms per group pass = {base_ms / G:.3f} ms per batch")
for n in ("step_prenet1_q_cq_fc", "step_attention_prenet2", "step_lstm_cell"):
    l, t = base_prof[n]
    print(f"   {n:28s} {l:5d} x {1e3 * t / l:7.2f} us (event-bracketed)")
for setting in opts:
    ms, prof, out = run(setting)
    print(f"{setting}: {ms:.3f} ms per group pass; outputs bit-identical to default: {torch.equal(out, base_out)}")
    for n in ("step_prenet1_q_cq_fc", "step_attention_prenet2", "step_lstm_cell"):
        l, t = prof[n]
        print(f"   {n:28s} {l:5d} x {1e3 * t / l:7.2f} us (event-bracketed)")
