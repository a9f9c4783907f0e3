"""Front-end conv kernel: the split-bf16 path in its forms (two output frames per block = option frontend_x3 2; the same with the next slab's staging
interleaved between the MFMA groups = 3; one frame per block = 1) vs the f32 MFMA path (0), time per batch and difference (2 and 3: the same bits).
-> profiles/rNN_frontend_forms.txt"""
import os, sys, time, torch
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
sd = synth.synth_state_dict()
tens = {k: v.cuda() for k, v in sd.items()}
def mk(x3):
    nm = native.NativeModel(); nm.set_option("frontend_x3", x3); nm.load(tens, list(sd.keys())); return nm
a, aq, a1, b = mk(2), mk(3), mk(1), mk(0)
for B in (32, 128, 256):
    v = synth.synth_video(32, 29, tag="bench").cuda().repeat(B // 32, 1, 1, 1, 1)
    oa, ob, oq = a.op_frontend(v), b.op_frontend(v), aq.op_frontend(v)
    print(f"B={B}: max|x3 - f32| = {(oa-ob).abs().max().item():.3e}  (|out| max {ob.abs().max().item():.2f}); interleaved staging bit-identical to the two-frame form: {torch.equal(oa, oq)}")
    if B == 32:
        v88 = v[:2, :, :7, 4:92, 4:92].contiguous()
        print(f"   88x88, T=7: interleaved staging bit-identical: {torch.equal(a.op_frontend(v88), aq.op_frontend(v88))}")
    for name, nm in (("x3, two frames per block", a), ("x3, two frames, interleaved staging", aq), ("x3, one frame per block ", a1), ("f32                     ", b)):
        for _ in range(3): nm.op_frontend(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): nm.op_frontend(v)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"  {name}: {dt*1e3:.3f} ms  ({2*1178.6e6*B/dt/1e12:.1f} TFLOP/s useful)")
