#!/usr/bin/env python
"""does a clip's K / V (119 KB, constant over the 300 steps) stay in its XCD's L2 between steps?  Clip r already runs on XCD
r mod 8 in every launch (block index = row, blocks go round-robin over the XCDs).  This runs N back-to-back launches of the attention kernel
ALONE (no LSTM weight stream in between) - run it under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` and compare the hit rate with the one
inside the real step (profiles/r03_pmc_decode.json): equal = the L2 does not keep lines across a kernel boundary; higher = the weight streams
evict them."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
B, T = int(os.environ.get("ROWS", "256")), 29
G = B // 32
v = synth.synth_video(32, T, tag="bench").cuda().repeat(G, 1, 1, 1, 1)
emb = synth.synth_speaker_embedding(32, tag="bench").cuda().repeat(G, 1)
gum = synth.synth_gumbel(32 * 4, tag="bench").cuda().repeat(G, 1)
state, _ = nm.decoder_prologue(native.build_visual(nm.encoder_fwd(v), emb), emb, gum)
ws = nm.workspace(B, T, 96, 96, 300, state.device)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
native.check(native.lib().l2s_op_step_attn_chain(nm._h, state.data_ptr(), B, T, 300, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
ev1.record(); torch.cuda.synchronize()
print(f"300 attention launches alone at {B} rows: {ev0.elapsed_time(ev1) / 300 * 1e3:.2f} us per launch")
