"""Is the training step bound by the host's enqueue rate or by the GPU?  Host time to enqueue N steps (no sync inside) next to the wall time
with a final synchronise, and the GPU-side span of one step from HIP events."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
from lip2speech_amd.training import model_forward_backward, draw_dropout
B, T, S = int(os.environ.get("B", 8)), 29, 77
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
bound = {k: v.clone().cuda() for k, v in sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point()}
is_buf = lambda k: k.endswith(("running_mean", "running_var", "pos_table"))
grads = {k: torch.zeros_like(v) for k, v in bound.items() if not is_buf(k)}
nm.train_bind(bound, grads); nm.train_set_bn(True, 0.1)
video = synth.synth_video(B, T, tag="tt").cuda(); emb = synth.synth_speaker_embedding(B, tag="tt").cuda()
gum = synth.synth_gumbel(B * 4, tag="tt").cuda(); mels = synth.synth_mels(B, S, tag="tt").cuda()
gate = torch.zeros(B, S, device="cuda"); gate[:, -1] = 1
drop = draw_dropout(B, T, S, "cuda")
def step(): return model_forward_backward(nm, video, emb, gum, mels, gate, drop=drop)
for _ in range(3): step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
th = time.perf_counter() - t0
torch.cuda.synchronize(); tw = time.perf_counter() - t0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); step(); e1.record(); torch.cuda.synchronize()
print(f"forward+backward, B={B}: host enqueue {th/n*1e3:.2f} ms/step, wall {tw/n*1e3:.2f} ms/step, one step alone between events {e0.elapsed_time(e1):.2f} ms")
# one step enqueued into an EMPTY queue: the host cannot be held back by a full queue here, so this is the host's own cost
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter() - t0; torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print(f"  one step into an empty queue: host returns after {t1*1e3:.2f} ms, GPU done after {t2*1e3:.2f} ms")
