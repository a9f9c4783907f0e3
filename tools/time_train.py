"""Time one training step (config 3 shape: B=8 clips per GPU, T=29, S=77 teacher targets) and its parts."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
from lip2speech_amd.training import model_forward_backward, FlatBuffer, AdamWAmsgrad

B, T, S = int(os.environ.get("B", 8)), 29, 77
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked", "pos_table"))
params = {k: v.cuda() for k, v in sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point() and not is_buf(k)}
grads = {k: torch.zeros_like(v) for k, v in params.items()}
nm.train_bind(params, grads)
video = synth.synth_video(B, T, tag="tt").cuda(); emb = synth.synth_speaker_embedding(B, tag="tt").cuda()
gum = synth.synth_gumbel(B * 4, tag="tt").cuda(); mels = synth.synth_mels(B, S, tag="tt").cuda()
gate = torch.zeros(B, S, device="cuda"); gate[:, -1] = 1

def step():
    return model_forward_backward(nm, video, emb, gum, mels, gate)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B} S={S}: forward+backward {dt*1e3:.2f} ms/step  ({B/dt:.1f} clips/s)")
native.profile_enable(True); native.profile_reset()
step(); torch.cuda.synchronize()
prof = sorted(native.profile_read(), key=lambda r: -r[2])
tot = sum(r[2] for r in prof)
print(f"profiled GPU time {tot:.2f} ms")
for name, cnt, ms in prof[:22]:
    print(f"  {name:40s} {cnt:6d} launches {ms:8.3f} ms  {100*ms/tot:5.1f}%")
native.profile_enable(False)
t0 = time.perf_counter()
nm.load({k: v for k, v in {**{k: v.cuda() for k, v in sd.items()}, **params}.items()}, list(sd.keys()))
torch.cuda.synchronize()
print(f"host re-pack of the weight blob: {(time.perf_counter()-t0)*1e3:.0f} ms")
