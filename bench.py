#!/usr/bin/env python
"""Headline benchmark of the hot path: mel-frames/s of Lip2Speech.inference on LRW-shaped clips.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: bench.py starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole path (visual encoder -> decoder prologue -> 300 autoregressive steps ->
post-net -> stop bookkeeping) over one synthetic batch of B=32 clips of 29 frames (BASELINE.json configs[1]).
Inputs are resident in HBM before the timed region.  With N ranks every rank runs its own B=32 batches (clips are
independent: weak scaling, no data-path collective); the reported value is the whole-job aggregate
N*K*B*S / max-over-ranks(time).

The K steps of a rank are independent batches.  One pass is a chain of ~1 500 dependent launches whose 1 200 step kernels are
latency-bound at 32 rows, so the steps are advanced `--group` (default 8) batches per launch chain (`l2s_inference_multi`: the G batches
are rows of the same launches, on one weight blob) with `--inflight` chains in flight (default: three - the step kernels' blocks take
half a compute unit, so launches of different chains run side by side on the CUs; `InflightPool.chains_for`), each on its own HIP
stream from its own host thread (lip2speech_amd.parallel.InflightPool).  Every step is still one full pass over one B=32 batch and every batch's
results are bit-identical to `l2s_inference` on it alone (tests/test_gpu_parity.py); `one_batch_at_a_time` in the JSON line is the same K
steps as K sequential `l2s_inference` calls, `one_chain_at_a_time` the same with one chain, `latency` what one group takes alone.
Each stream needs a hardware queue: GPU_MAX_HW_QUEUES is raised to 8 below, before the HIP runtime starts (the secondary figures
use four chains).

The JSON line also carries
  roofline     for the kernel with the largest share of GPU time (HIP events per launch in a separate profiled pass, l2s_profile_*),
               `frac` = the FLOPs one launch executes / the duration of ONE launch alone on its stream (one event pair around a chain of
               launches, in the block form the timed region launches) / the ceiling of the pipe it runs on; `frac_overlapped` beside it =
               what the chip spends per launch with the timed region's chains in flight; `traffic` from the committed PMC passes;
  cpu_baseline the CPU oracle (oracle/l2s_oracle.py, "port" of the reference math, verified against the imported
               reference) timed on this host's cores on the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # must precede the first HIP call (see the docstring)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lip2speech_amd import native, synth  # noqa: E402

B, T, HW, S = 32, 29, 96, 300
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* = fp32 vector rate
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak (same guide); the split-bf16 kernels issue SIX bf16 products per fp32 product
HBM_PEAK_GBS = 8000.0


def kernel_model(name, rows=B):
    """Algorithmic (FLOPs, bytes) of ONE launch of a kernel of the path over `rows` clips (G batches of B=32 per launch chain), T=29:
    SURVEY.md §8(d)'s per-clip figures x the clips one launch processes; weights counted once per launch."""
    m = native.min_T(T)
    w4 = 4
    R = rows
    table = {
        # both decoder LSTM layers run the same kernel (600 launches per pass); figures are the per-launch mean of the two layers.
        # ALGORITHMIC work (decoder.py:423): each layer is [x | h] (K = 512 + 512) against 2048 gate rows, and the
        # 512 -> 256 attention_proj of decoder.py:420 that the phase-merged layer 0 absorbs is credited to it.  What the kernel EXECUTES is
        # more (layer 0 runs K = 1280: the u columns of W_ih twice, once for the prenet output and once for o = a @ V'): `executed` below, reported under roofline.frac_executed - never as frac.
        "step_lstm_cell": (2 * R * (2048 * 1024 + 2048 * 1024 + 256 * 512) / 2,
                           ((2048 * (1024 + 1024) / 2 + 256 * 512 / 2 + 2048) * w4 + R * (1024 + 3 * 512) * w4)),
        "step_prenet1_q_cq_fc": (2 * R * (256 * 512 + 512 * 1024 + 256 * 1024 + 81 * 512),
                                 (256 * 512 + 512 * 1024 + 256 * 1024 + 96 * 512 + 1024) * w4 + R * (2048 + 256 + 512 + 256 + 81) * w4),
        "step_attention_prenet2": (2 * R * (2 * T * 512 + 2 * m * 256 + 256 * 256),
                                   R * (2 * T * 512 + 2 * m * 256 + 512 + 256 + 512 + 256 + T) * w4 + (256 * 256) * w4),
        "step_attention_proj": (2 * R * 256 * 512, 256 * 512 * w4 + R * (512 + 512) * w4),
        "step_fc_out_stop": (2 * R * 81 * 512, 96 * 512 * w4 + R * (512 + 160) * w4),
        "frontend3d_conv_bn_prelu_pool": (2 * 1178.6e6 * R, (R * 3 * T * HW * HW + R * T * 24 * 24 * 24 + 17640) * w4),
        "postnet_conv_gemm": (2 * 4.34e6 * S * R / 5, (R * S * (80 + 512 * 4 * 2 + 80) + 4.35e6) * w4 / 5),
    }
    # fused ShuffleNet units (shufflenetv2.py:42-104), one launch over R*T frames: the unit's MACs (two / three 1x1 convs + 3x3 depthwise) and its
    # map read once + written once + its weights
    NF = R * T
    for tag, hh, half in (("h12", 144, 58), ("h6", 36, 116), ("h3", 9, 232)):       # stride-1: x2 -> pw -> dw -> pw, x1 passes through
        table["shuffle_unit_s1_fused_" + tag] = (2 * NF * hh * (2 * half * half + 9 * half), (NF * hh * 2 * half * 2 + 2 * half * half + 15 * half) * w4)
    for tag, h, cin, half in (("st2", 24, 24, 58), ("st3", 12, 116, 116), ("st4", 6, 232, 232)):      # stride-2: (dw s2 -> pw) and (pw -> dw s2 -> pw)
        ho = h // 2
        table["shuffle_unit_s2_fused_" + tag] = (2 * NF * (ho * ho * (9 * cin + cin * half + 9 * half + half * half) + h * h * cin * half),
                                                 (NF * (h * h * cin + ho * ho * 2 * half) + 2 * cin * half + half * half + 9 * cin + 15 * half) * w4)
    return table.get(name)


# which matrix pipe a kernel of the path runs on: "bf16x3" = the bf16 matrix cores through the exact three-way split (six
# bf16 MFMAs per fp32 product: ceiling 2 500 / 6 = 416.7 TFLOP/s fp32-equivalent); "f32" = v_mfma_f32_*_f32 (157.3 TFLOP/s)
KERNEL_PIPE = {"step_lstm_cell": "bf16x3", "postnet_conv_gemm": "bf16x3", "frontend3d_conv_bn_prelu_pool": "bf16x3",
               "step_prenet1_q_cq_fc": "f32", "step_fc_out_stop": "f32", "step_attention_proj": "f32",
               "shuffle_unit_s1_fused_h12": "bf16x3", "shuffle_unit_s1_fused_h6": "bf16x3", "shuffle_unit_s1_fused_h3": "bf16x3", "shuffle_unit_s2_fused_st3": "bf16x3",
               "shuffle_unit_s2_fused_st2": "f32", "shuffle_unit_s2_fused_st4": "f32"}


def mfma_roof(name, flops, avg_s):
    """Pipe-aware MFMA roofline entry: `frac` is against the ceiling of the pipe the kernel RUNS on - for the split-bf16 kernels
    executed bf16 FLOPs (6 x algorithmic) / duration / 2.5 PFLOP/s, i.e. algorithmic / 416.7 TFLOP/s - and can never exceed 1;
    the ratio to the fp32 matrix peak (the data's dtype) is kept as `frac_fp32_equiv`."""
    ach = flops / avg_s / 1e12
    if KERNEL_PIPE.get(name) == "bf16x3":
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0
        return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "frac_fp32_equiv": ach / FP32_MFMA_PEAK_TFLOPS,
                "pipe": "bf16 matrix cores, exact three-way split: 6 bf16 MFMA products per fp32 product; peak = 2500 / 6 TFLOP/s fp32-equivalent",
                "executed_bf16_tflops": 6 * ach}
    return {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "pipe": "f32 matrix instructions (v_mfma_f32_*_f32)"}


def executed_flops(name, rows=B):
    """FLOPs the launch really issues where they differ from the algorithmic count (phase-merged weights)."""
    return {"step_lstm_cell": 2 * rows * 2048 * (1024 + 1024) / 2}.get(name)      # both layers run K = 1024 (layer 0 on [content | prenet + a.V' | h0], option hoist_vproj = 2)


def cpu_baseline(seconds_budget=12.0):
    """The CPU oracle on this host's cores, same B=32 / T=29 / S=300 workload, same synthetic checkpoint.
    torch's intra-op threading over-subscribes badly on the tiny per-step ops when given every hardware thread of a
    big host, so a short probe (B=32, S=20) picks the best thread count first; `cores` is the count actually used."""
    from oracle import l2s_oracle as orc
    sd = synth.synth_state_dict()
    video = synth.synth_video(B, T, tag="bench")
    emb = synth.synth_speaker_embedding(B, tag="bench")
    gum = synth.synth_gumbel(B * native.min_T(T), tag="bench")
    hw = os.cpu_count() or 1
    cands = sorted({c for c in (64, 32, 16, 8) if c <= hw} or {hw}, reverse=True)   # all 256 SMT threads: 100x slower, not probed
    probe = {}
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            orc.inference(sd, video, emb, gum, S=4)
            best = None
            for _ in range(2):                                 # best of two S=60 passes per candidate
                t0 = time.time()
                orc.inference(sd, video, emb, gum, S=60)
                dt = time.time() - t0
                best = dt if best is None else min(best, dt)
            probe[c] = best
        # the short probe only RANKS the candidates (the 300-step loop weighs the per-step ops more than a 60-step pass does: a probe alone picked 8, 16
        # or 32 threads from run to run): the two best are both measured on the real workload and the faster one is reported
        finals = {}
        warm = 0.0
        for c in sorted(probe, key=probe.get)[:2]:
            torch.set_num_threads(c)
            t0 = time.time()
            orc.inference(sd, video, emb, gum, S=S)         # warm-up pass (also page-in)
            warm += time.time() - t0
            ts = []
            while sum(ts) < seconds_budget / 2 and len(ts) < 3:
                t0 = time.time()
                orc.inference(sd, video, emb, gum, S=S)
                ts.append(time.time() - t0)
            finals[c] = sorted(ts)
        cores = min(finals, key=lambda c: finals[c][len(finals[c]) // 2])
        times = finals[cores]
    med = times[len(times) // 2]
    return {"value": B * S / med, "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full passes of the B={B},T={T},S={S} batch after a warm-up pass; median {med:.2f}s; the two thread counts the probe ranks "
                      f"best are both measured this way (medians {({k: round(v[len(v) // 2], 2) for k, v in finals.items()})}), the faster is reported; probe (best of two S=60 passes, seconds): {({k: round(v, 2) for k, v in probe.items()})} of {hw} hardware threads"}


def cpu_baseline_train(Bt, St, seconds_budget=25.0):
    """One training step of the same shape on this host's cores: forward (batch-statistics BatchNorm) -> 4-term loss -> autograd backward
    through the CPU oracle -> clip -> torch AdamW(amsgrad); `cores` from the inference probe's best setting."""
    from oracle import l2s_oracle as orc
    sd = synth.synth_state_dict()
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked", "pos_table"))      # noqa: E731
    keys = [k for k in sd if k.startswith(("encoder.", "decoder."))]
    work = {k: (sd[k].clone().requires_grad_(sd[k].is_floating_point() and not is_buf(k))) for k in keys}
    params = [work[k] for k in keys if work[k].requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-6, amsgrad=True)
    video = synth.synth_video(Bt, T, tag="train0")
    emb = synth.synth_speaker_embedding(Bt, tag="train0")
    gum = synth.synth_gumbel(Bt * native.min_T(T), tag="train0")
    mels = synth.synth_mels(Bt, St, tag="train0")
    gate = torch.zeros(Bt, St)
    gate[:, -1] = 1.0
    hw = os.cpu_count() or 1
    cores = min(32, hw)
    torch.set_num_threads(cores)
    times = []
    while (sum(times) < seconds_budget and len(times) < 4) or len(times) < 2:
        t0 = time.time()
        with orc.batch_statistics():
            outs = orc.forward_eval(work, video, emb, mels, gum)
        terms = orc.loss_terms(outs, mels, gate)
        opt.zero_grad()
        terms[-1].backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        times.append(time.time() - t0)
    best = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": Bt / best, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{len(times) - 1} training steps after 1 warm-up of the B={Bt}, T={T}, S={St} batch on the CPU oracle with torch autograd "
                      f"(batch-statistics BatchNorm, no dropout, no teacher forcing); median {best:.2f} s per step"}


def train_kernel_model(name, Bt, St):
    """Algorithmic (FLOPs, bytes) of ONE launch of the training step's kernels that have a closed form (literal 6-phase step)."""
    w4 = 4
    table = {
        "train_step_lstm_cell": (2 * Bt * 2048 * 1024, (2048 * 1024 + 2048) * w4 + Bt * (1024 + 3 * 512 + 4 * 2048) * w4),
        "train_bwd_lstm_dx": (2 * Bt * 1024 * 2048, (1024 * 2048) * w4 + Bt * (2048 + 1024) * w4),       # d[x|h] = dgates (B,2048) x [W_ih | W_hh]
    }
    return table.get(name)


def train_setup(bf16, rank=0, world=1):
    """Model, optimizer, reducer and inputs of the training bench; returns (step, ctx).  One step = `Lip2Speech.forward` (encoder + decoder,
    S=77 targets) + 4-term loss + backward through everything + bucketed gradient all-reduce (RCCL, N>1) + global-norm clip + fused
    AdamW(amsgrad) + device-side re-pack of the weight blob, on B=8 clips per GPU, train() semantics."""
    from model.model import get_network
    from lip2speech_amd.training import AdamWAmsgrad, GradAllReducer, draw_dropout, model_forward_backward
    Bt, St = 8, 77
    net = get_network("train").cuda()
    net.load_state_dict({k: v for k, v in synth.synth_state_dict().items() if k.startswith(("encoder.", "decoder."))}, strict=False)
    flat = net._train_state()
    nm = net.native_model()
    opt = AdamWAmsgrad(flat, lr=1e-4, weight_decay=1e-6)
    reducer = GradAllReducer(flat.grad)
    n_dec = reducer.buckets_covering(sum(p.numel() for p in net.decoder.parameters()))     # the flat buffer holds the decoder group first
    tag = f"train{rank}"
    video = synth.synth_video(Bt, T, tag=tag).cuda()
    emb = synth.synth_speaker_embedding(Bt, tag=tag).cuda()
    gum = synth.synth_gumbel(Bt * native.min_T(T), tag=tag).cuda()
    mels = synth.synth_mels(Bt, St, tag=tag).cuda()
    gate = torch.zeros(Bt, St, device="cuda")
    gate[:, -1] = 1.0
    bos = dict(net.decoder.named_parameters())["BOS"]
    mask = torch.zeros(St, dtype=torch.bool)
    mask[1::2] = True                                   # half of the steps teacher-forced (tf_ratio 0.5 regime)

    nm.train_set_bn(True, 0.1)
    nm.set_option("train_bf16", 1 if bf16 else 0)

    def step():
        drop = draw_dropout(Bt, T, St, video.device)
        # the decoder's buckets are reduced while the encoder backward still runs; the rest follows
        out = model_forward_backward(nm, video, emb, gum, mels, gate, teacher_mask=mask, bos=bos.detach(), drop=drop,
                                     on_decoder_grads=lambda: reducer.start(0, n_dec))
        reducer.start(n_dec)
        mul = reducer.wait()
        opt.step(max_norm=1.0, grad_mul=mul)
        nm.train_refresh_weights()
        return out

    return step, {"Bt": Bt, "St": St, "net": net, "nm": nm, "opt": opt, "reducer": reducer}


def train_leg(steps=5, warmup=3):
    """`train_step_B8` of the default line (BASELINE.json configs[2] at its per-GPU shape, fp32): `warmup` untimed + `steps` timed training
    steps between two device synchronisations, exactly the step `bench.py --mode train` times (train.py:150-193)."""
    step, ctx = train_setup(False)
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = out["loss"].cpu()
    assert torch.isfinite(loss).all(), "non-finite training loss"
    return {"ms_per_step": dt / steps * 1e3, "value": ctx["Bt"] * steps / dt, "unit": "clips/s", "steps": steps, "warmup": warmup, "batch_per_gpu": ctx["Bt"],
            "decode_steps": ctx["St"], "dtype": "f32", "final_loss": float(loss[4]),
            "note": "one data-parallel training step per step at B=8 per GPU, T=29, S=77 (SURVEY.md section 8(d) config 3): forward with tapes + 4-term loss + "
                    "backward + gradient bucketing (all-reduce is a no-op at one rank) + global-norm clip + fused AdamW-amsgrad + device re-pack of the "
                    "weight blob; train() semantics (batch-statistics BatchNorm, five dropout sites, half the steps teacher-forced); the full line: "
                    "`python bench.py --mode train`"}


def train_main(args):
    """Secondary bench line: training throughput (see train_setup), fp32 or --bf16 operands, N ranks."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("L2S_BENCH_ONE_DEVICE"):      # test hook: several ranks on ONE GPU (with L2S_BENCH_BACKEND=gloo) to exercise the N>1 code path
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("L2S_BENCH_BACKEND", "nccl"))
    step, ctx = train_setup(args.bf16, rank, world)
    Bt, St = ctx["Bt"], ctx["St"]

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = out["loss"].cpu()
    assert torch.isfinite(loss).all(), "non-finite loss"
    # per-kernel HIP-event timing of ONE more step (events around every launch perturb the pipeline: separate pass); every rank takes the
    # step (it contains the gradient all-reduce), rank 0 records it
    if rank == 0:
        native.profile_enable(True)
        native.profile_reset()
    step()
    torch.cuda.synchronize()
    if rank == 0:
        prof = sorted(native.profile_read(), key=lambda r: -r[2])
        native.profile_enable(False)
        gpu_ms = sum(r[2] for r in prof)
        roof = {"kernels_by_gpu_time": [{"kernel": n, "launches": l, "total_ms": ms, "avg_us_event_per_launch": ms / l * 1e3, "share": ms / gpu_ms}
                                        for n, l, ms in prof[:8]], "gpu_ms_event_sum": gpu_ms}
        for n, l, ms in prof:
            km = train_kernel_model(n, Bt, St)
            if km:
                flops, nbytes = km
                avg_s = ms / l * 1e-3
                roof.update(kernel=n, launches_per_step=l, avg_us=avg_s * 1e6, bound="hbm", achieved=nbytes / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=nbytes / avg_s / 1e9 / HBM_PEAK_GBS, algorithmic_flops=flops, algorithmic_bytes=nbytes, traffic=None,
                            note="largest kernel with a closed-form cost: at 8 rows per launch the step kernels stream their weights once per "
                                 "launch (AI = 4 FLOP/B): per-launch HIP-event brackets add ~1.8 us to these us-scale kernels")
                break
        cpu = cpu_baseline_train(Bt, St) if (world == 1 and not args.skip_cpu_baseline) else None
        print(json.dumps({
            "metric": "training clips/sec (forward + loss + backward + all-reduce + clip + AdamW-amsgrad + weight re-pack)",
            "value": world * Bt * args.steps / elapsed, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 operands / f32 accumulate in encoder, prologue and post-net GEMMs; f32 loop, statistics, master weights" if args.bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": "LRW training step, batch=8 per GPU, 29x96x96 clips, S=77 mel targets, half of the steps teacher-forced, "
                                   "38.4 M parameters, train() semantics (batch-statistics BatchNorm, dropout)", "batch_per_gpu": Bt, "frames": T,
                       "decode_steps": St, "parallelism": f"dp{world} (one bucketed gradient all-reduce of 153.7 MB per step)"},
            "roofline": roof, "cpu_baseline": cpu,
            "speedup_vs_cpu_baseline": (world * Bt * args.steps / elapsed) / cpu["value"] if cpu else None,
            "final_loss": float(loss[4])}), flush=True)
    if dist:
        dist.destroy_process_group()


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run and let rank 0 print
    the line.  Fails loudly when fewer than N devices are visible instead of quietly measuring one."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible to this process; refusing to report an {args.gpus}-GPU figure")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 192 inference passes (0.9 s) / 20 training steps")
    ap.add_argument("--warmup", type=int, default=None, help="default: 16 / 3")
    ap.add_argument("--skip-cpu-baseline", action="store_true", help="for profiler runs")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="run-time option of the library (include/l2s.h) set as the process "
                    "default before any model is created - A/B runs; the line records them under config.options")
    ap.add_argument("--skip-train-leg", action="store_true", help="inference mode: leave the train_step_B8 leg out of the line (profiler runs)")
    ap.add_argument("--group", type=int, default=8, help="independent B=32 batches advanced per launch chain (l2s_inference_multi, 1..8); 1 = one batch per chain")
    ap.add_argument("--inflight", type=int, default=None, help="launch chains in flight per GPU (HIP streams + host threads); default: InflightPool.chains_for(steps, group) "
                    "- two, or three where K steps cut into fuller launches that way (20 steps: 7 + 7 + 6 instead of 4 x 5); 1 = strictly sequential chains "
                    "(measured at 8 batches per chain: 1 chain 4.78 ms per batch, 2 chains 4.30, 3 chains 4.22-4.33; their relative phase does not matter)")
    ap.add_argument("--bf16", action="store_true", help="--mode train: bf16 operands in the GEMMs / Conv1d stacks of encoder, prologue and post-net "
                    "(option train_bf16; BASELINE.json configs[2] names bf16), fp32 accumulation / master weights / recurrent loop")
    ap.add_argument("--mode", choices=["inference", "train"], default="inference",
                    help="inference = the headline metric (default); train = one data-parallel training step per 'step' (SURVEY.md §8 config 3)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.mode == "train" else 192
    if args.warmup is None:
        args.warmup = 3 if args.mode == "train" else 16
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args)
    if args.mode == "train":
        return train_main(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if os.environ.get("L2S_BENCH_ONE_DEVICE"):      # test hook: several ranks on ONE GPU (with L2S_BENCH_BACKEND=gloo) to exercise the N>1 code path
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("L2S_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend=backend)

    # replicated weights, per-rank shard of clips (no exchange step on the inference path)
    from lip2speech_amd.parallel import InflightPool
    for kv in args.opt:
        if kv.split("=")[0] in native.DIAG_OPTIONS and native.lib() is not native.diag():
            raise SystemExit(f"bench.py --opt {kv}: a block-form A/B switch of the diagnostic build (include/l2s_diag.h) - run with L2S_LIB=diag (tools/ab_bench.sh does)")
        native.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    sd = synth.synth_state_dict()
    tensors = {k: v.cuda() for k, v in sd.items()}
    G = max(1, min(8, args.group))
    NI = max(1, args.inflight) if args.inflight is not None else InflightPool.chains_for(args.steps, G)
    pool = InflightPool(tensors, list(sd.keys()), n_inflight=NI, group=G)
    nm = pool.model
    # every slot of every chain gets its own batch (different clips, embeddings and noise): concurrent passes share nothing but the weights
    n_distinct = G * NI
    batches = []
    for i in range(n_distinct):
        tag = f"bench{rank}.{i}" if (rank or i) else "bench"
        batches.append((synth.synth_video(B, T, tag=tag).cuda(), synth.synth_speaker_embedding(B, tag=tag).cuda(),
                        synth.synth_gumbel(B * native.min_T(T), tag=tag).cuda()))
    video, emb, gum = batches[0]
    work = lambda n: [batches[i % n_distinct] for i in range(n)]      # noqa: E731

    def timed(run):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, res

    # W untimed warm-up steps (at least one group per chain), then - still untimed - windows of one round of groups until two consecutive
    # windows agree within 2 % (clocks, allocator and the chains' relative phase have settled), at most 8 windows
    pool.map(work(max(args.warmup, n_distinct)), S=S)
    extra, prev_w = 0, None
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pool.map(work(n_distinct), S=S)
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        extra += n_distinct
        if prev_w is not None and abs(w - prev_w) <= 0.02 * max(w, prev_w):
            break
        prev_w = w
    if args.steps % n_distinct:
        # K is not whole rounds of full groups: the timed region will run balanced groups of fewer rows per launch (20 steps = 4 groups of 5)
        # - one untimed pass of that very grouping, so that launch shape is not met for the first time inside the timed region
        pool.map(work(min(args.steps, 4 * n_distinct)), S=S)
        extra += min(args.steps, 4 * n_distinct)
    elapsed, outs = timed(lambda: pool.map(work(args.steps), S=S))   # EXACTLY K steps
    assert all(torch.isfinite(o[0]).all() for o in outs), "non-finite mel output"
    # reference figures, same K steps: strictly one batch at a time on one stream; and one chain of G batches at a time
    seq_elapsed, _ = timed(lambda: [nm.inference(*batches[i % n_distinct], S=S) for i in range(args.steps)])
    one_chain = InflightPool(model=nm, n_inflight=1, group=G)
    chain_elapsed, _ = timed(lambda: one_chain.map(work(args.steps), S=S))
    # latency of ONE group through the path with the chip otherwise idle (what a caller waits for G batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nm.inference_multi(batches[:G], S=S) if G > 1 else nm.inference(*batches[0], S=S)
    torch.cuda.synchronize()
    group_latency = time.perf_counter() - t0

    # the evaluate.py path: forward(tf_ratio=1) with S = 77 target frames per clip - the same grouping and chains as
    # `value`, through the entry the reference-facing callers use (InflightPool.imap -> l2s_forward_eval_multi)
    S77 = 77
    fwd_job = lambda b: {"entry": "forward", "video": b[0], "emb": b[1], "gumbel": b[2], "S": S77}      # noqa: E731
    list(pool.imap(work(2 * n_distinct), fwd_job))
    fwd_elapsed, _ = timed(lambda: list(pool.imap(work(args.steps), fwd_job)))
    # secondary figures on a pool of four single-batch chains (the entry points below are per batch)
    pool4 = pool if (G == 1 and NI == 4) else InflightPool(model=nm, n_inflight=4, group=1)
    # the boundary handed HOST buffers (pinned): every step first copies its 102.6 MB of frames, the speaker embedding and the Gumbel
    # noise to the GPU on its own stream, overlapping the other chains' compute.  Never `value` (inputs resident there).
    host_batches = [tuple(t.cpu().pin_memory() for t in batches[i % n_distinct]) for i in range(4)]
    h2d = lambda model, b: model.inference(*(t.cuda(non_blocking=True) for t in b), S=S)      # noqa: E731
    pool4.map(host_batches, fn=h2d)
    h2d_elapsed, _ = timed(lambda: pool4.map([host_batches[i % 4] for i in range(args.steps)], fn=h2d))
    # the same with the data boundary on the device: the decoded uint8 frames (25.7 MB per batch) cross PCIe packed and
    # l2s_normalise_pad_frames builds the fp32 batch (datasets.device.PackedFrames; bit-identical to the host collate)
    from lip2speech_amd.datasets import PackedFrames
    gen = torch.Generator().manual_seed(1234)
    u8_batches = [(PackedFrames([torch.randint(0, 256, (T, HW, HW, 3), dtype=torch.uint8, generator=gen) for _ in range(B)]), hb[1], hb[2]) for hb in host_batches]
    u8 = lambda b: (b[0].to_device(), b[1].cuda(non_blocking=True), b[2].cuda(non_blocking=True))      # noqa: E731
    u8_shape = lambda b: (B, 3, T, HW, HW)      # noqa: E731
    pool.map(u8_batches * (n_distinct // 4 + 1), S=S, prepare=u8, shape_of=u8_shape)
    u8_elapsed, _ = timed(lambda: pool.map([u8_batches[i % 4] for i in range(args.steps)], S=S, prepare=u8, shape_of=u8_shape))

    # the bf16 leg (BASELINE.json configs[4] names bf16): a model of its own with option "infer_bf16" - front-end conv, GEMMs and Conv1d
    # stacks with bf16 operands, fp32 accumulation, fp32 recurrent loop - same steps, grouping and chains as `value`.  Never `value`: its
    # mel frames sit inside a stated band of the reference's (tests: mean |d| < 2e-2, max < 0.15), not inside the 1e-3 fp32 gate.
    nm16 = native.NativeModel()
    nm16.set_option("infer_bf16", 1)
    nm16.load(tensors, list(sd.keys()))
    pool16 = InflightPool(model=nm16, n_inflight=NI, group=G)
    pool16.map(work(2 * n_distinct), S=S)
    bf16_elapsed, outs16 = timed(lambda: pool16.map(work(args.steps), S=S))
    bf16_dev = max(float((a[0] - b[0]).abs().mean()) for a, b in zip(outs16[:n_distinct], outs[:n_distinct]))

    # BASELINE.json configs[0] / demo.py:60-90: what ONE caller waits for a small batch on an otherwise idle GPU (B = 2 and B = 1, S = 300):
    # median wall time of 7 calls after 3 warm-ups, each call bracketed by device synchronisations
    def latency_ms(nb, persist):
        a = (video[:nb].contiguous(), emb[:nb].contiguous(), gum[:nb * native.min_T(T)].contiguous())
        nm.set_option("persist_decode", persist)
        ts = []
        try:
            for i in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nm.inference(*a, S=S)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        finally:
            nm.set_option("persist_decode", 4)
        return sorted(ts[3:])[3] * 1e3
    lat = {nb: (latency_ms(nb, 4), latency_ms(nb, 0)) for nb in (2, 1)} if world == 1 else None
    train = train_leg() if (world == 1 and not args.skip_train_leg) else None

    if rank == 0:
        # per-kernel HIP-event timing in its own pass over ONE group (events around every launch perturb the pipeline)
        native.profile_enable(True)
        native.profile_reset()
        for _ in range(2):
            nm.inference_multi(batches[:G], S=S) if G > 1 else nm.inference(*batches[0], S=S)
        torch.cuda.synchronize()
        prof = sorted(native.profile_read(), key=lambda r: -r[2])
        native.profile_enable(False)
        gpu_ms = sum(r[2] for r in prof)
        name, launches, total_ms = prof[0]
        avg_s = total_ms / launches * 1e-3
        rows = G * B
        roof = {"kernel": name, "rows_per_launch": rows, "launches_per_group_pass": launches // 2, "share_of_gpu_time": total_ms / gpu_ms,
                "avg_us_event_per_launch": avg_s * 1e6}
        if name == "step_lstm_cell":
            # per-launch event brackets inflate a us-scale kernel by ~1.8 us; time the same launches as one chain between ONE event pair on the
            # launch stream (l2s_op_lstm_cell_chain of the diagnostic build libl2s_diag.so: the product's kernel sources, same launch code).
            # The block form is the one the timed region launches: with NI >= 2 chains in flight the half-CU form (skinny_rc4h), else the eight-wave one.
            native.set_thread_chains(NI)
            try:
                avg_s = nm.lstm_cell_chain_us(rows, 300) * 1e-6
            finally:
                native.set_thread_chains(1)
            roof["timing"] = ("one HIP-event pair around a chain of 616 launches (308 x {layer 0, layer 1}; 16 warm-up) ALONE on the launch stream, in the block form "
                              "the timed region launches (half-CU blocks when chains overlap): the duration rocprofv3 reports for the kernel")
            roof["block_form"] = "skinny_rc4h_kernel<4,2,{6,2},3> (four waves, half a compute unit)" if NI >= 2 else "skinny_rc8x_kernel<4,2,{6,2},4> (eight waves)"
        roof["avg_us"] = avg_s * 1e6
        overlapped_s = None
        if name == "step_lstm_cell" and NI > 1:
            # Secondary figure: the timed region keeps NI launch chains in flight and the step kernels' blocks take half a compute unit, so LSTM launches
            # of different chains run side by side on the CUs: what the chip spends per launch is the wall time of NI such chains at once / (NI x launches)
            # (the in-kernel block-stamp log confirms the overlap without the host's clock: profiles/r06_overlap_stamps.txt)
            import threading
            def chains_at_once(n_pairs=300):
                bar = threading.Barrier(NI + 1)
                def work(i):
                    torch.cuda.set_device(local_rank)
                    native.set_thread_chains(NI)
                    with torch.cuda.stream(pool.streams[i]):
                        bar.wait()
                        nm.lstm_cell_chain_us(rows, n_pairs)
                        pool.streams[i].synchronize()
                th = [threading.Thread(target=work, args=(i,)) for i in range(NI)]
                for t_ in th: t_.start()
                torch.cuda.synchronize()
                bar.wait()
                t0_ = time.perf_counter()
                for t_ in th: t_.join()
                return (time.perf_counter() - t0_) / (NI * (2 * n_pairs + 16))      # the chain op runs 8 warm-up pairs first
            chains_at_once(50)
            overlapped_s = min(chains_at_once() for _ in range(3))
        model = kernel_model(name, rows)
        if model:
            flops, nbytes = model
            ex = executed_flops(name, rows)
            if ex:                       # the roofline is priced on what the launch EXECUTES (the algorithmic count credits work hoisted out of the loop)
                roof["algorithmic_flops_survey"] = flops
                flops = ex
            ai = flops / nbytes
            if ai < FP32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
                ach = nbytes / avg_s / 1e9
                roof.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)
            else:
                roof.update(mfma_roof(name, flops, avg_s))
            roof["executed_flops"] = flops
            roof["algorithmic_bytes"] = nbytes
            roof["arithmetic_intensity"] = ai
            if overlapped_s and roof["bound"] == "mfma":
                roof["chains_at_once"] = NI
                roof["avg_us_overlapped"] = overlapped_s * 1e6
                roof["frac_overlapped"] = roof["frac"] * avg_s / overlapped_s
            if roof["bound"] == "mfma":
                roof["note"] = ("frac = EXECUTED FLOPs of one launch (256 rows x 2048 gate rows x K = 1024, x2) / the duration of ONE launch alone on its stream / the "
                                "ceiling of the pipe the kernel runs on (bf16 matrix cores, six products per fp32 product: 416.7 TFLOP/s fp32-equivalent) - the "
                                "definition of rounds 1-4 (0.238 in round 4), frozen; frac_fp32_equiv: the same rate against the fp32 matrix peak of 157.3 TFLOP/s; "
                                "frac_overlapped / avg_us_overlapped: wall time of the chains the timed region keeps in flight / all their launches - what the chip "
                                "spends per launch when launches of different chains share the CUs (a utilisation figure, not a kernel duration)")
        # bytes per launch at the L2's memory side from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), collected OFFLINE on the
        # block forms of the timed region with its chains in flight (tools/profile_r6.sh, one counter group per pass) and committed under profiles/
        roof["traffic"] = None
        try:
            pmc_file = next(f for f in ("r06_pmc_decode_half3.json", "r05_pmc_decode.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            k = pmc["kernels"].get(name)
            if k and pmc.get("rows_per_launch") == rows:
                roof["traffic"] = k["traffic_bytes_per_launch"]
                roof["traffic_kernel_symbols"] = k.get("kernel_symbols")
                roof["traffic_source"] = f"offline: profiles/{pmc_file} (rocprofv3 --pmc on the symbols listed, {rows} rows per launch), not measured in this run"
                roof["traffic_over_algorithmic_bytes"] = k["traffic_bytes_per_launch"] / roof["algorithmic_bytes"] if roof.get("algorithmic_bytes") else None
                roof["l2_hit_rate_offline"] = k.get("l2_hit_rate")
                for key in ("avg_us_rocprofv3_one_chain", "avg_us_rocprofv3"):
                    if k.get(key):
                        roof[key + "_offline"] = k[key]
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        # the other kernels with a closed-form cost model, same pass (per-launch HIP-event brackets: us-scale kernels carry ~1.8 us of it)
        others = []
        for oname, olaunches, oms in prof[1:]:
            om = kernel_model(oname, rows)
            if not om or len(others) >= 11:
                continue
            oavg = oms / olaunches * 1e-3
            oflops, obytes = om
            if oflops / obytes < FP32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
                o = {"bound": "hbm", "achieved": obytes / oavg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
                o["frac"] = o["achieved"] / o["peak"]
            else:
                o = mfma_roof(oname, oflops, oavg)
            o.update(kernel=oname, launches_per_group_pass=olaunches // 2, share_of_gpu_time=oms / gpu_ms, avg_us_event_per_launch=oavg * 1e6)
            others.append(o)
        roof["other_kernels"] = others
        # whole-path figure against the fp32 matrix peak (36.18 MFLOP per mel frame)
        per_gpu = B * S * args.steps / elapsed
        roof["path_tflops"] = per_gpu * 36.18e6 / 1e12
        roof["path_frac_fp32_peak"] = roof["path_tflops"] / FP32_MFMA_PEAK_TFLOPS      # mixed pipes (f32 trunk / first phase, split-bf16 elsewhere): a ratio to the data's dtype peak, not a pipe roofline

        value = world * B * S * args.steps / elapsed
        line = {
            "metric": "mel-frames/sec at LRW batch=32, 29-frame clips (Lip2Speech.inference, S=300)",
            "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LRW single-word, batch=32 per step, 29x96x96 RGB mouth crops, S=300 decode steps, speaker embedding "
                                   "supplied (encoding=voice), random-init weights; every step is one full pass over one B=32 batch",
                       "batch_per_step": B, "frames": T, "decode_steps": S, "options": list(args.opt), "library": os.path.basename(native.LIB_PATH), "parallelism": f"dp{world} (clip sharding, no collective)",
                       "batches_per_launch_chain": G, "launch_chains_in_flight_per_gpu": NI, "distinct_batches_per_gpu": n_distinct,
                       "collective_backend": backend, "ranks": world},
            "warmup_extra_steps_until_steady": extra,
            "latency": {"ms_one_group_alone": group_latency * 1e3, "batches_in_group": G,
                        "note": "wall time of ONE launch chain over G batches on an otherwise idle GPU: what a caller waits for its G results; "
                                "ms_per_step above is elapsed / steps with all chains busy, a throughput figure, not a latency"},
            "one_chain_at_a_time": {"value": world * B * S * args.steps / chain_elapsed, "ms_per_step": chain_elapsed / args.steps * 1e3,
                                    "note": f"the same K steps, groups of {G} issued one after the other on one stream"},
            "one_batch_at_a_time": {"value": world * B * S * args.steps / seq_elapsed, "ms_per_step": seq_elapsed / args.steps * 1e3,
                                    "note": "the same K steps as K calls of l2s_inference, strictly sequential (one B=32 batch per launch chain)"},
            "evaluate_forward_S77": {"value": world * B * S77 * args.steps / fwd_elapsed, "unit": "mel-frames/s", "ms_per_step": fwd_elapsed / args.steps * 1e3,
                                     "tflops_algorithmic": world * B * S77 * args.steps / fwd_elapsed * 85.28e6 / 1e12,
                                     "note": "Lip2Speech.forward(tf_ratio=1) in eval mode, S=77 (evaluate.py:38), through the callers' streaming entry "
                                             "(InflightPool.imap -> l2s_forward_eval_multi), same batches per chain and chains in flight as `value`"},
            "host_resident_inputs": {"value": world * B * S * args.steps / h2d_elapsed, "unit": "mel-frames/s", "ms_per_step": h2d_elapsed / args.steps * 1e3,
                                     "note": "PCIe-inclusive: each step copies its batch from pinned host memory on its own stream first; four single-batch chains in flight"},
            "bf16_leg": {"value": world * B * S * args.steps / bf16_elapsed, "unit": "mel-frames/s", "ms_per_step": bf16_elapsed / args.steps * 1e3,
                         "mean_abs_mel_deviation_from_fp32_path": bf16_dev,
                         "note": "model option infer_bf16: front-end conv, GEMMs and Conv1d stacks with bf16 operands (fp32 accumulation), recurrent loops, "
                                 "fused ShuffleNet units and statistics fp32; same steps, grouping and chains as `value`; a different precision - never `value`"},
            "host_resident_uint8_frames": {"value": world * B * S * args.steps / u8_elapsed, "unit": "mel-frames/s", "ms_per_step": u8_elapsed / args.steps * 1e3,
                                           "note": "PCIe-inclusive with the data boundary on the device: every step's packed uint8 frames (25.7 MB per batch) are copied from "
                                                   "pinned host memory and normalised + padded by l2s_normalise_pad_frames one group ahead on the pool's copy stream; same grouping and chains as `value`"},
            "roofline": roof,
        }
        if lat:
            for nb, (ms, ms_launch) in lat.items():
                line[f"latency_B{nb}_S300"] = {"ms": ms, "value": nb * S / ms * 1e3, "unit": "mel-frames/s", "ms_launch_per_phase_loop": ms_launch,
                                               "note": f"one l2s_inference call on B={nb} clips (T=29, S=300) alone on the GPU, call to results; median of 7; "
                                                       "the library's default for one or two clips: the decode loop as ONE persistent weight-stationary launch "
                                                       "(pdecode.hip, option persist_decode); ms_launch_per_phase_loop = the same call with four launches per step"}
        if train:
            line["train_step_B8"] = train
        if world == 1 and not args.skip_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
