"""Do the kernels of several launch chains overlap on the chip when a block takes half a compute unit?

Two measurements, each for the default block forms (every step kernel's blocks sit alone on their CU: "lstm_x3" = 2, "flat_half" = 0, "attn_lds" = 1: round 4's) and the half-CU
forms ("lstm_x3" = 3: four-wave LSTM blocks of <= 256 registers; "flat_half" = 1: four-wave first-phase blocks of <= 153 registers):
  * LSTM launches alone: n chains of 600 launches (l2s_op_lstm_cell_chain, 256 rows) at once on n streams;
  * the whole decode loop: n chains of l2s_decode_steps (S = 300, G x 32 rows each, prepared state) at once.
Reported: wall time from a common start to the last chain's end / (launches or steps of ONE chain) = what a chain waits per launch / step, and
that divided by n = what the chip spends per launch / step.  One model, one host thread and one HIP stream per chain.
Usage: python tools/coresident_probe.py [G]   -> profiles/r05_coresident_probe.txt"""
import os, sys, threading, time
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_amd import native, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROWS, T, S = 32 * G, 29, 300
N = int(os.environ.get("PAIRS", "300"))
REP = int(os.environ.get("REP", "3"))
CHAINS = [int(c) for c in os.environ.get("CHAINS", "1,2,3,4").split(",")]      # e.g. CHAINS=3 MODE=decode FORMS=half under rocprofv3
MODE = os.environ.get("MODE", "lstm,decode").split(",")
FORMS = os.environ.get("FORMS", "default,half").split(",")
sd = synth.synth_state_dict()
tensors = {k: v.cuda() for k, v in sd.items()}


def together(n, fn):
    """fn(i) on n threads / streams from a common start; wall seconds until all are done (GPU included)."""
    streams = [torch.cuda.Stream() for _ in range(n)]
    bar = threading.Barrier(n + 1)

    def work(i):
        torch.cuda.set_device(0)
        native.set_thread_chains(HINT)   # 1: the blocks of a chain that has the chip to itself (round 4's forms), 2: the half-CU forms - whatever n, also for the one-chain column
        with torch.cuda.stream(streams[i]):
            bar.wait()
            fn(i)
            streams[i].synchronize()

    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th: t.start()
    torch.cuda.synchronize()
    bar.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    return time.perf_counter() - t0


HINT = 1
for name, opts in (("default blocks ", {"lstm_x3": 2, "flat_half": 0, "attn_lds": 1}), ("half-CU blocks ", {"lstm_x3": 3, "flat_half": 2, "attn_lds": 2})):
    if name.split("-")[0].split()[0] not in FORMS:
        continue
    HINT = 1 if name.startswith("default") else 2
    nm = native.NativeModel()
    nm.set_option("persist_decode", 0)
    nm.set_option("use_graph", 0)
    for k, v in opts.items():
        nm.set_option(k, v)
    for kv in os.environ.get("L2S_OPT", "").split(","):      # extra options on top, e.g. L2S_OPT=half_min_mts=8
        if kv:
            nm.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    nm.load(tensors, list(sd.keys()))
    # ---- LSTM launches alone
    nm.lstm_cell_chain_us(ROWS, 20)
    line = f"{name} LSTM launches, {ROWS} rows:"
    for n in (CHAINS if "lstm" in MODE else []):
        w = min(together(n, lambda i: nm.lstm_cell_chain_us(ROWS, N)) for _ in range(REP))
        per = w * 1e6 / (2 * N + 16)
        line += f"  {n} chain(s): {per:6.2f} us per launch of a chain, {per / n:5.2f} for the chip;"
    if "lstm" in MODE:
        print(line, flush=True)
    if "decode" not in MODE and "attn" not in MODE:
        continue
    # ---- the whole decode loop
    states = []
    for i in range(4):
        v = synth.synth_video(ROWS, T, tag=f"cp{i}").cuda(); e = synth.synth_speaker_embedding(ROWS, tag=f"cp{i}").cuda(); g = synth.synth_gumbel(ROWS * 4, tag=f"cp{i}").cuda()
        feat = nm.encoder_fwd(v)
        st, _ = nm.decoder_prologue(native.build_visual(feat, e), e, g)
        states.append(st)
        del v, feat
    torch.cuda.synchronize()
    if "attn" in MODE:      # the attention launch alone (l2s_op_step_attn_chain: same K / V / content state every launch, zero queries)
        L = native.lib()
        wss = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(4)]
        NA = 2 * N
        def attn_chain(i):
            native.check(L.l2s_op_step_attn_chain(nm._h, states[i].data_ptr(), ROWS, T, NA, wss[i].data_ptr(), wss[i].numel(), torch.cuda.current_stream().cuda_stream))
        attn_chain(0); torch.cuda.synchronize()
        line = f"{name} attention launches, {ROWS} rows:"
        for n in CHAINS:
            w = min(together(n, attn_chain) for _ in range(REP))
            per = w * 1e6 / NA
            line += f"  {n} chain(s): {per:6.2f} us per launch of a chain, {per / n:5.2f} for the chip;"
        print(line, flush=True)
    nm.decode_steps(states[0], ROWS, T, S, want_attn=False)
    line = f"{name} decode loop,   {ROWS} rows:"
    for n in (CHAINS if "decode" in MODE else []):
        w = min(together(n, lambda i: nm.decode_steps(states[i], ROWS, T, S, want_attn=False)) for _ in range(REP))
        per = w * 1e6 / S
        line += f"  {n} chain(s): {per:6.2f} us per step of a chain, {per / n:5.2f} for the chip;"
    if "decode" in MODE:
        print(line, flush=True)
