#!/usr/bin/env python
"""Do the step kernels of several launch chains really run side by side?  An overlap proof that depends neither on the host's clock nor on a profiler:
the diagnostic build's block-stamp log (include/l2s_diag.h l2s_op_stamp_log) - thread 0 of EVERY block of the decode step's kernels (LSTM launches, first
phase, attention) records {entry, exit} on the chip's 100 MHz constant clock (s_memrealtime) while n chains run at once on n streams, exactly the
launches of bench.py's timed region (half-CU block forms, 256 rows).  From the block records: a launch = the blocks of one kernel kind and one chain that
start within 3 us of each other; its span = first entry .. last exit.  Reported per kernel kind: launches, mean span of a launch, and - over the window
in which all chains are active - the UNION of the spans (time during which at least one launch of that kind is on the chip) per launch = what the chip
spends on a launch of that kind when the chains overlap, and the mean number of launches in flight.  Usage: CHAINS=3 MODE=lstm|decode python
tools/overlap_stamps.py [G]   -> profiles/r06_overlap_stamps.txt"""
import os, sys, threading
os.environ.setdefault("L2S_LIB", "diag")      # tools run on the diagnostic build (libl2s_diag.so: product ABI + include/l2s_diag.h)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lip2speech_amd import native, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ROWS, T = 32 * G, 29
S = int(os.environ.get("S", "150"))
N = int(os.environ.get("PAIRS", "150"))
KIND = {1: "LSTM launch (skinny_rc4h / rc8x)", 2: "first phase (skinny_flat)", 3: "attention + prenet2 (step_attn)"}
sd = synth.synth_state_dict()
nm = native.NativeModel()
nm.set_option("persist_decode", 0)
for kv in filter(None, os.environ.get("L2S_OPT", "").split(",")):
    nm.set_option(kv.split("=")[0], int(kv.split("=")[1]))
nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
D = native.diag()
CAP = 6_000_000
log = torch.zeros(1 + 3 * CAP, dtype=torch.int64, device="cuda")


def together(n, hint, fn):
    streams = [torch.cuda.Stream() for _ in range(n)]
    bar = threading.Barrier(n + 1)

    def work(i):
        torch.cuda.set_device(0)
        native.set_thread_chains(hint)
        with torch.cuda.stream(streams[i]):
            bar.wait()
            fn(i)
            streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th: t.start()
    torch.cuda.synchronize()
    bar.wait()
    for t in th: t.join()


def union(iv):
    """total length of the union of [a, b) intervals (ticks), and the sum of their lengths"""
    iv = iv[np.argsort(iv[:, 0])]
    tot, cur_a, cur_b = 0, iv[0, 0], iv[0, 1]
    for a, b in iv[1:]:
        if a > cur_b:
            tot += cur_b - cur_a; cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    return tot + cur_b - cur_a, int((iv[:, 1] - iv[:, 0]).sum())


def analyse(title, n):
    torch.cuda.synchronize()
    cnt = int(log[0].item())
    assert 0 < cnt <= CAP, f"stamp log: {cnt} records (capacity {CAP})"
    rec = log[1:1 + 3 * cnt].view(cnt, 3).cpu().numpy()
    t0, t1, tag = rec[:, 0], rec[:, 1], rec[:, 2]
    kind = (tag >> 60) & 0xF
    key = tag & ((1 << 60) - 1)
    print(f"== {title}: {cnt} block records, {n} chain(s), tick = 10 ns")
    launches = {}      # kind -> array of (start, end, chain)
    for k in sorted(set(kind.tolist())):
        spans = []
        for ci, c in enumerate(sorted(set(key[kind == k].tolist()))):
            m = (kind == k) & (key == c)
            o = np.argsort(t0[m]); a, b = t0[m][o], t1[m][o]
            # launches of one chain and kind never overlap (a dependent chain) and all have the same grid: first guess the launches by a gap of > 3 us
            # between block entries, take the most common size as the grid, then cut the entry-ordered records into runs of exactly that many blocks
            cut = np.flatnonzero(np.diff(a) > 300) + 1
            sizes = np.diff(np.r_[0, cut, len(a)])
            nb = int(np.bincount(sizes).argmax())
            if len(a) % nb == 0:
                cut = np.arange(nb, len(a), nb)
            for lo, hi in zip(np.r_[0, cut], np.r_[cut, len(a)]):
                spans.append((a[lo], b[lo:hi].max(), ci, hi - lo))
        launches[k] = np.array(spans, dtype=np.int64)
    # the window in which every chain is active: from the latest first launch to the earliest last launch over the chains
    allsp = np.concatenate([v for v in launches.values()])
    chains = sorted(set(allsp[:, 2].tolist()))
    if n > 1 and len(chains) < n:
        print(f"   (only {len(chains)} distinct operand keys for {n} chains: launches of different chains share a key - per-chain grouping by the 3 us gap rule only)")
    lo = max(allsp[allsp[:, 2] == c][:, 0].min() for c in chains); hi = min(allsp[allsp[:, 2] == c][:, 1].max() for c in chains)
    if hi <= lo:
        print(f"   the chains did not overlap in time (latest first launch {lo} > earliest last launch {hi}): whole record instead")
        lo, hi = allsp[:, 0].min(), allsp[:, 1].max()
    win = allsp[(allsp[:, 0] >= lo) & (allsp[:, 1] <= hi)]
    u_all, s_all = union(win[:, :2])
    print(f"   window with all chains active: {(hi - lo) / 100:.1f} us, {len(win)} launches of the step kernels; at least one on the chip {100.0 * u_all / (hi - lo):.1f} % of it; "
          f"mean launches in flight while busy {s_all / u_all:.2f}; chip time per launch (union / launches) {u_all / len(win) / 100:.2f} us")
    for k, sp in launches.items():
        w = sp[(sp[:, 0] >= lo) & (sp[:, 1] <= hi)]
        if not len(w):
            continue
        u, s_ = union(w[:, :2])
        d = (w[:, 1] - w[:, 0]) / 100.0
        print(f"   {KIND.get(k, k):34s} {len(w):5d} launches, {int(np.median(w[:, 3])):4d} blocks each: span of a launch mean {d.mean():6.2f} us (median {np.median(d):6.2f}, p90 {np.percentile(d, 90):6.2f}); "
              f"union of spans / launches = {u / len(w) / 100:5.2f} us per launch for the chip; {s_ / u:4.2f} launches of this kind in flight while one is")


MODE = os.environ.get("MODE", "lstm,decode").split(",")
CH = [int(c) for c in os.environ.get("CHAINS", "1,3").split(",")]
states = []
if "decode" in MODE:
    for i in range(max(CH)):
        v = synth.synth_video(ROWS, T, tag=f"cp{i}").cuda(); e = synth.synth_speaker_embedding(ROWS, tag=f"cp{i}").cuda(); g = synth.synth_gumbel(ROWS * 4, tag=f"cp{i}").cuda()
        st, _ = nm.decoder_prologue(native.build_visual(nm.encoder_fwd(v), e), e, g)
        states.append(st)
        del v
    nm.decode_steps(states[0], ROWS, T, 8, want_attn=False)
nm.lstm_cell_chain_us(ROWS, 10)
torch.cuda.synchronize()
for n in CH:
    hint = 2 if n > 1 or os.environ.get("FORMS", "half") == "half" else 1
    if "lstm" in MODE:
        together(n, hint, lambda i: nm.lstm_cell_chain_us(ROWS, 20))      # warm-up: the threads' workspaces are allocated, the block forms loaded
        log[0] = 0
        native.check(D.l2s_op_stamp_log(log.data_ptr(), CAP), D)
        together(n, hint, lambda i: nm.lstm_cell_chain_us(ROWS, N))
        native.check(D.l2s_op_stamp_log(None, 0), D)
        analyse(f"LSTM launches alone, {ROWS} rows, {2 * N + 16} launches per chain (l2s_op_lstm_cell_chain), chains hint {hint}", n)
    if "decode" in MODE:
        together(n, hint, lambda i: nm.decode_steps(states[i], ROWS, T, 10, want_attn=False))
        log[0] = 0
        native.check(D.l2s_op_stamp_log(log.data_ptr(), CAP), D)
        together(n, hint, lambda i: nm.decode_steps(states[i], ROWS, T, S, want_attn=False))
        native.check(D.l2s_op_stamp_log(None, 0), D)
        analyse(f"decode loop, {ROWS} rows, S = {S} steps per chain (l2s_decode_steps), chains hint {hint}", n)
