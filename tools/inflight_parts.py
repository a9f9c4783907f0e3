"""Throughput of the parts of a pass with n batches in flight: decode loop only, dense part only (encoder + prologue + post-net)."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native, synth
B, T, S = 32, 29, 300
sd = synth.synth_state_dict()
NT = int(os.environ.get("NT", 3))
ctx = []
for i in range(NT):
    nm = native.NativeModel(); nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
    v = synth.synth_video(B, T, tag=f"b{i}").cuda(); e = synth.synth_speaker_embedding(B, tag=f"b{i}").cuda(); g = synth.synth_gumbel(B * 4, tag=f"b{i}").cuda()
    feat = nm.encoder_fwd(v); vis = native.build_visual(feat, e); state, _ = nm.decoder_prologue(vis, e, g)
    mel, stop, _ = nm.decode_steps(state, B, T, S)
    ctx.append(dict(nm=nm, v=v, e=e, g=g, state=state, mel=mel, st=torch.cuda.Stream()))
def decode(c): c["nm"].decode_steps(c["state"], B, T, S)
def dense(c):
    nm = c["nm"]; feat = nm.encoder_fwd(c["v"]); vis = native.build_visual(feat, c["e"]); nm.decoder_prologue(vis, c["e"], c["g"]); nm.postnet(c["mel"])
def run(fn, n, par):
    def worker(c):
        with torch.cuda.stream(c["st"]):
            for _ in range(n): fn(c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if par:
        th = [threading.Thread(target=worker, args=(c,)) for c in ctx]
        for t in th: t.start()
        for t in th: t.join()
    else:
        for c in ctx: worker(c)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (n * len(ctx)) * 1e3
for name, fn in (("decode loop", decode), ("dense part (encoder + prologue + post-net)", dense)):
    run(fn, 2, True)
    print(f"{name:45s}: one at a time {run(fn, 6, False):6.2f} ms per batch, {NT} in flight {run(fn, 6, True):6.2f} ms per batch")
