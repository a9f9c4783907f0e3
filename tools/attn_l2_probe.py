#!/usr/bin/env python
"""300 back-to-back launches of the step's attention kernel ALONE at ROWS clips (env, default 256) - under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum`
(tools/attn_l2_sweep.sh) this tells whether a clip's keys / values survive in its XCD's L2 from one launch to the next: at 256 rows the footprint
(3.7-4.1 MB per XCD) is the L2's size, at 128 / 64 rows it is a half / a quarter of it.
-> profiles/rNN_attn_l2_probe.txt"""
import os, sys, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B, T = int(os.environ.get("ROWS", "256")), 29
G = max(1, B // 32)
v = synth.synth_video(32, T, tag="bench").cuda().repeat(G, 1, 1, 1, 1)[:B]
emb = synth.synth_speaker_embedding(32, tag="bench").cuda().repeat(G, 1)[:B]
gum = synth.synth_gumbel(32 * 4, tag="bench").cuda().repeat(G, 1)[:B * 4]
state, _ = nm.decoder_prologue(native.build_visual(nm.encoder_fwd(v), emb), emb, gum)
ws = nm.workspace(B, T, 96, 96, 300, state.device)
L = native.lib()
def chain(n): native.check(L.l2s_op_step_attn_chain(nm._h, state.data_ptr(), B, T, n, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
chain(20); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); chain(300); e1.record(); torch.cuda.synchronize()
print(f"{B} rows: {e0.elapsed_time(e1) / 300 * 1e3:.2f} us per attention launch (300 back to back, alone)")
