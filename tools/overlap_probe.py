"""What two launch chains in flight can overlap: the 300 decode steps of one chain (latency-bound launches of 64-256 blocks) against the
dense stages of another (encoder + post-net, thousands of blocks per launch), each looped on its own HIP stream from its own host thread -
alone, together, and together with the decode stream at high priority.  Usage: python tools/overlap_probe.py [rows]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_amd import native, synth

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, S = 29, 300
sd = synth.synth_state_dict()
tensors = {k: v.cuda() for k, v in sd.items()}
nm = native.NativeModel()
nm.load(tensors, list(sd.keys()))
video = torch.cat([synth.synth_video(32, T, tag=f"b{i}") for i in range(ROWS // 32)]).cuda()
emb = torch.cat([synth.synth_speaker_embedding(32, tag=f"b{i}") for i in range(ROWS // 32)]).cuda()
gum = torch.cat([synth.synth_gumbel(32 * 4, tag=f"b{i}") for i in range(ROWS // 32)]).cuda()
feat = nm.encoder_fwd(video)
vis = native.build_visual(feat, emb)
state0, _ = nm.decoder_prologue(vis, emb, gum, want_dis=False)
mel0, _, _ = nm.decode_steps(state0.clone(), ROWS, T, S, want_attn=False)
torch.cuda.synchronize()


def decode_loop(stream, n, out, gate):
    with torch.cuda.stream(stream):
        st = state0.clone()
        nm.decode_steps(st, ROWS, T, S, want_attn=False)
        stream.synchronize()
        gate.wait()
        t0 = time.perf_counter()
        out["decode_t0"] = t0
        for _ in range(n):
            nm.decode_steps(st, ROWS, T, S, want_attn=False)
        stream.synchronize()
        out["decode_t1"] = time.perf_counter()
        out["decode"] = (time.perf_counter() - t0) / n * 1e3


def dense_loop(stream, n, out, gate):
    with torch.cuda.stream(stream):
        nm.encoder_fwd(video); nm.postnet(mel0)
        stream.synchronize()
        gate.wait()
        t0 = time.perf_counter()
        out["dense_t0"] = t0
        for _ in range(n):
            nm.encoder_fwd(video)
            nm.postnet(mel0)
        stream.synchronize()
        out["dense_t1"] = time.perf_counter()
        out["dense"] = (time.perf_counter() - t0) / n * 1e3


import ctypes
_hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """a HIP stream confined to the CUs whose bit is set (bit i = CU i / 8 of XCD i % 8 on this part: the driver deals the mask round-robin over the XCDs)"""
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    h = ctypes.c_void_p()
    rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


def run(label, pri_decode, pri_dense, n=6, both=True, which=("decode", "dense"), split=0):
    if split:
        sd_, sn_ = masked_stream(set(range(split))), masked_stream(set(range(split, 256)))
    else:
        sd_, sn_ = torch.cuda.Stream(priority=pri_decode), torch.cuda.Stream(priority=pri_dense)
    out = {}
    ths = []
    gate = threading.Barrier(len(which))
    if "decode" in which:
        ths.append(threading.Thread(target=decode_loop, args=(sd_, n, out, gate)))
    if "dense" in which:
        ths.append(threading.Thread(target=dense_loop, args=(sn_, n, out, gate)))
    for t in ths: t.start()
    for t in ths: t.join()
    t0 = min(v for k, v in out.items() if k.endswith("_t0"))
    wall = (max(v for k, v in out.items() if k.endswith("_t1")) - t0) * 1e3
    spans = "  ".join(f"{k} [{(out[k + '_t0'] - t0) * 1e3:6.1f}, {(out[k + '_t1'] - t0) * 1e3:6.1f}]" for k in which)
    print(f"{label:44s} " + "  ".join(f"{k} {out[k]:7.2f} ms/pass" for k in which) + f"   wall {wall:7.1f} ms for {n} passes each   {spans}")
    return out


lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print(f"rows {ROWS}: stream priority range {lo}..{hi}")
a = run("decode alone", 0, 0, which=("decode",))
b = run("dense (encoder + post-net) alone", 0, 0, which=("dense",))
print(f"   serial sum {a['decode'] + b['dense']:.2f} ms")
run("together, equal priority", 0, 0)
run("together, decode stream high priority", -1, 0)
run("together, dense stream high priority", 0, -1)
run("together, equal priority (repeat)", 0, 0)
for c in (32, 48, 64, 96, 128):
    print(f"--- decode confined to {c} CUs ({c // 8} per XCD), dense to the other {256 - c}")
    run(f"decode alone on {c} CUs", 0, 0, which=("decode",), split=c)
    run(f"dense alone on {256 - c} CUs", 0, 0, which=("dense",), split=c)
    run(f"together, {c} | {256 - c}", 0, 0, split=c)
