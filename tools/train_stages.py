"""GPU time of every stage of one training step (forward + backward, B=8, T=29, S=77, train() semantics): CUDA events around each
native entry point of lip2speech_amd.training.model_forward_backward, averaged over N steps.
-> profiles/rNN_train_stages.txt"""
import os, sys, collections, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth, training
B, T, S = int(os.environ.get("B", 8)), 29, 77
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
for kv in filter(None, os.environ.get("L2S_OPTS", "").split(",")):
    k, v = kv.split("="); nm.set_option(k, int(v))
bound = {k: v.clone().cuda() for k, v in sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point()}
is_buf = lambda k: k.endswith(("running_mean", "running_var", "pos_table"))
grads = {k: torch.zeros_like(v) for k, v in bound.items() if not is_buf(k)}
nm.train_bind(bound, grads); nm.train_set_bn(True, 0.1)
video = synth.synth_video(B, T, tag="tt").cuda(); emb = synth.synth_speaker_embedding(B, tag="tt").cuda()
gum = synth.synth_gumbel(B * 4, tag="tt").cuda(); mels = synth.synth_mels(B, S, tag="tt").cuda()
gate = torch.zeros(B, S, device="cuda"); gate[:, -1] = 1
drop = training.draw_dropout(B, T, S, "cuda")
mask = torch.zeros(S, dtype=torch.bool); mask[::2] = True
bos = bound["decoder.BOS"]
spans = collections.defaultdict(list)
def wrap(obj, name):
    fn = getattr(obj, name)
    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record(); spans[name].append((e0, e1)); return r
    setattr(obj, name, timed)
for n in ("train_encoder_fwd", "train_prologue_fwd", "train_steps_fwd", "train_postnet_fwd", "train_pack_weights", "train_postnet_bwd", "train_steps_bwd",
          "train_prologue_bwd", "train_encoder_bwd"): wrap(nm, n)
wrap(training, "loss_terms")
def step(): return training.model_forward_backward(nm, video, emb, gum, mels, gate, teacher_mask=mask, bos=bos, drop=drop)
for _ in range(3): step()
torch.cuda.synchronize(); spans.clear()
N = 8
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): step()
e1.record(); torch.cuda.synchronize()
tot = 0.0
for n, ev in spans.items():
    ms = sum(a.elapsed_time(b) for a, b in ev) / N; tot += ms
    print(f"  {n:24s} {ms:7.3f} ms")
print(f"  sum {tot:.2f} ms; whole step between events {e0.elapsed_time(e1)/N:.2f} ms")
