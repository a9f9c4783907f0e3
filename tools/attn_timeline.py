#!/usr/bin/env python
"""Timeline of the attention blocks of the step's second launch (step_attn_kernel) from the stamped build (`l2s_op_attn_timeline`): thread 0 of every
attention block stamps the 100 MHz wall clock at entry / requests issued / q visible / logits computed / after the barrier / weights visible / stored.
ROWS env (default 256): clips per launch; the launch runs alone, 40 times back to back, the last one's stamps are read.
-> profiles/rNN_attn_timeline.txt"""
import os, sys, torch, numpy as np
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B, T = int(os.environ.get("ROWS", "256")), 29
G = B // 32
v = synth.synth_video(32, T, tag="bench").cuda().repeat(G, 1, 1, 1, 1)
emb = synth.synth_speaker_embedding(32, tag="bench").cuda().repeat(G, 1)
gum = synth.synth_gumbel(32 * 4, tag="bench").cuda().repeat(G, 1)
state, _ = nm.decoder_prologue(native.build_visual(nm.encoder_fwd(v), emb), emb, gum)
ws = nm.workspace(B, T, 96, 96, 300, state.device)
L = native.lib()
def chain(n): native.check(L.l2s_op_step_attn_chain(nm._h, state.data_ptr(), B, T, n, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
chain(20); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); chain(300); e1.record(); torch.cuda.synchronize()
print(f"{B} rows: production launch {e0.elapsed_time(e1) / 300 * 1e3:.2f} us (300 back to back, alone)")
ts = torch.zeros(B * 8, dtype=torch.int64, device="cuda")
native.check(L.l2s_op_attn_timeline(ts.data_ptr()))
chain(40); torch.cuda.synchronize()
native.check(L.l2s_op_attn_timeline(None))
t = ts.cpu().numpy().reshape(B, 8)[:, :7].astype(np.float64) * 0.01
t -= t[:, 0].min()
names = ["entry", "requests issued", "q visible (barrier 1)", "logits computed", "after barrier 2", "weights visible (barrier 3)", "stored, drained"]
print(f"{'stamp':28s} {'min':>6s} {'median':>7s} {'max':>6s}   (us since the first attention block entered)")
for i, n in enumerate(names): print(f"{n:28s} {t[:, i].min():6.2f} {np.median(t[:, i]):7.2f} {t[:, i].max():6.2f}")
d = np.diff(t, axis=1)
print("per-block phase durations (median us): " + " | ".join(f"{n}: {np.median(d[:, i]):.2f}" for i, n in enumerate(names[1:])))
print(f"attention blocks: lifetime median {np.median(t[:, 6] - t[:, 0]):.2f} us; span of all of them {t[:, 6].max():.2f} us")
