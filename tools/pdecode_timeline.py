#!/usr/bin/env python
"""Phase timeline of the persistent decode loop (pdecode.hip, option "persist_decode"): thread 0 of all 256 workgroups stamps 14 points of ONE step
(STEP env, default 150) of an l2s_decode_steps call on B clips (B env, default 1; T=29, S=300).
-> profiles/r04_pdecode_timeline.txt"""
import os, sys, torch, numpy as np
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
B = int(os.environ.get("B", 1)); step = int(os.environ.get("STEP", 150)); S = int(os.environ.get("S", 300)); T = 29
sd = synth.synth_state_dict()
nm = native.NativeModel(); nm.set_option("persist_decode", 8); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
v = synth.synth_video(B, T, tag=f"lat{B}").cuda(); e = synth.synth_speaker_embedding(B, tag=f"lat{B}").cuda(); g = synth.synth_gumbel(B * native.min_T(T), tag=f"lat{B}").cuda()
feat = nm.encoder_fwd(v); vis = native.build_visual(feat, e); state, _ = nm.decoder_prologue(vis, e, g)
for _ in range(2): nm.decode_steps(state, B, T, S)
L = native.lib()
names = ["step start", "P1 granules in", "P1 reduced", "P1 published", "P2 granules in", "P2 prenet2 reduced", "P2 logits", "P2 published",
         "P3 granules in", "P3 reduced", "P3 published", "P4 granules in", "P4 reduced", "P4 published"]
rows = []
for st in ([step] if S > step else [S // 2]):
    ts = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
    native.check(L.l2s_op_pdecode_timeline(ts.data_ptr(), st))
    nm.decode_steps(state, B, T, S); torch.cuda.synchronize()
    native.check(L.l2s_op_pdecode_timeline(None, 0))
    t = ts.cpu().numpy().reshape(256, 16)[:, :14].astype(np.float64) * 0.01
    t = t[t[:, 13] > 0]                                   # the workgroups of this launch (128 or 256)
    print(f"{len(t)} workgroups")
    t -= t[:, 0].min()
    print(f"persistent decode loop, B={B}, T={T}, S={S}: step {st}, us since the first workgroup entered the step (thread 0 of every workgroup)")
    print(f"{'stamp':22s} {'min':>7s} {'median':>7s} {'max':>7s}")
    for i, n in enumerate(names): print(f"{n:22s} {t[:, i].min():7.2f} {np.median(t[:, i]):7.2f} {t[:, i].max():7.2f}")
    d = np.diff(t, axis=1)
    print("median us per segment:", " | ".join(f"{names[i + 1]}: {np.median(d[:, i]):.2f}" for i in range(13)))
    print(f"step length (workgroup 0, start -> P4 published): {t[0, 13] - t[0, 0]:.2f} us")
