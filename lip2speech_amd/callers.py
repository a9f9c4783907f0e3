"""The model-facing halves of the reference's caller loops, without their I/O:

  * ``demo_clips`` / ``demo_clip`` - demo.py:60-90: per clip, speaker embedding from the VOICE tower (``--encoding voice``) or a
                         supplied one, ``net.inference(..., return_attention_map=True)``, truncate to ``output_lengths[0]``.
  * ``evaluate_mels`` / ``evaluate_net`` - evaluate.py:22-51: ``net(..., tf_ratio=1)[1]`` in eval mode over collated batches.
  * ``train_iterations`` - train.py:150-193.

The inference-side loops run on the GROUPED path by default: the loader's batches are prefetched ``group`` at a time into one launch chain
(``l2s_inference_multi`` / ``l2s_forward_eval_multi``) with ``n_inflight`` chains on the GPU (``Lip2Speech.inference_many`` /
``forward_many`` over ``parallel.InflightPool.imap``); results come back in loader order, each bit-identical to the single-batch call.

The reference then vocodes the mels (InverseMelScale + Griffin-Lim, torchaudio) and scores ESTOI (pystoi); both are
third-party, stochastic and out of scope here - these functions return the mels.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import native


def demo_clip(net, batch, speaker_encoder=None, speaker_embedding: Optional[torch.Tensor] = None, device="cuda"):
    """``batch`` = one item of ``DataLoader(ds, batch_size=1, collate_fn=test_collate_fn_pad)`` (one iteration of demo.py:60-90): the direct
    ``net.inference`` call, which for one clip takes the library's latency form (the decode loop as one persistent launch, option "persist_decode") -
    a single clip has no grouping to stay consistent with; ``demo_clips`` streams a whole loader through the grouped path instead."""
    if speaker_embedding is None and speaker_encoder is None:
        raise ValueError("pass a SpeakerEncoder (voice route) or a speaker_embedding")
    (videos, _), (audios, _), _, face_crops, _ = batch
    with torch.no_grad():
        emb = speaker_embedding if speaker_embedding is not None else speaker_encoder.inference(audios.to(device, non_blocking=True))
        mel, lengths, attn = net.inference(videos.to(device, non_blocking=True), face_crops, speaker_embedding=emb, return_attention_map=True)
    n = int(lengths[0])                                  # synchronises: a timed-out persistent launch is reported here, not handed on as NaN
    native.check_persist_timeouts()
    return mel[:1, :, :n], lengths, attn[:, :n]


def demo_clips(net, batches: Iterable, speaker_encoder=None, speaker_embedding: Optional[torch.Tensor] = None, device="cuda",
               group: int = 8, n_inflight: int = 3):
    """demo.py:60-90 over a whole loader: per clip the speaker embedding from the VOICE tower (``--encoding voice``) or a supplied one,
    ``net.inference(..., return_attention_map=True)``, truncation to ``output_lengths[0]``.  The clips are advanced ``group`` per launch
    chain with ``n_inflight`` chains on the GPU (``Lip2Speech.inference_many``); yields ``(mel, lengths, attention)`` per clip, in order."""
    if speaker_embedding is None and speaker_encoder is None:
        raise ValueError("pass a SpeakerEncoder (voice route) or a speaker_embedding")

    def calls():
        for (videos, _), (audios, _), _, face_crops, _ in batches:
            with torch.no_grad():
                emb = speaker_embedding if speaker_embedding is not None else speaker_encoder.inference(audios.to(device, non_blocking=True))
            yield videos, face_crops, emb, True

    for mel, lengths, attn in net.inference_many(calls(), group=group, n_inflight=n_inflight):
        n = int(lengths[0])
        yield mel[:1, :, :n], lengths, attn[:, :n]


def _evaluate_outputs(net, batches: Iterable, speaker_encoder, device, group: int, n_inflight: int):
    """The eval-mode ``net(..., tf_ratio=1)`` of evaluate.py:32-38 for every collated batch (``train_collate_fn_pad`` layout), on the grouped
    path: yields ``(batch, outputs)`` in loader order."""
    kept = []

    def calls():
        for batch in batches:
            (videos, vlen), (audios, alen), (melspecs, mlen, _gate), face_crops = batch
            kept.append(batch)
            with torch.no_grad():
                emb = speaker_encoder.inference(audios.to(device, non_blocking=True)) if speaker_encoder is not None else None
            yield videos, face_crops, audios, melspecs, vlen, alen, mlen, 1, {"speaker_embedding": emb}

    for out in net.forward_many(calls(), group=group, n_inflight=n_inflight):
        yield kept.pop(0), out


def evaluate_mels(net, batches: Iterable, speaker_encoder=None, device="cuda", group: int = 8, n_inflight: int = 3) -> List[torch.Tensor]:
    """Post-net mels of ``net(..., tf_ratio=1)[1]`` for every collated batch (``train_collate_fn_pad`` layout; evaluate.py:32-38), ``group``
    loader batches per launch chain (``l2s_forward_eval_multi``), ``n_inflight`` chains in flight."""
    was_training = net.training
    net.eval()
    try:
        return [out[1] for _, out in _evaluate_outputs(net, batches, speaker_encoder, device, group, n_inflight)]
    finally:
        net.train(was_training)


def evaluate_net(net, batches: Iterable, speaker_encoder=None, device="cuda", max_iters: int = 256, sampling_rate: int = None,
                 group: int = 8, n_inflight: int = 3, timings: Optional[dict] = None, vocoder_backend: str = "auto", metric: str = "auto") -> float:
    """Mean ESTOI of the vocoded predictions against the ground-truth audio (reference: evaluate.py:22-51): `net(..., tf_ratio=1)[1]`
    -> `MelSpec2Audio` (InverseMelScale + Griffin-Lim, `max_iters` each) -> `stoi(gt, pred, fs, extended=True)` per clip.  Vocoder and
    metric are restatements of third-party algorithms (parity unpinned); the mels come from the HIP path, `group` loader batches per launch
    chain, and the next groups' chains run while this thread vocodes and scores.  `timings` (a dict) receives the wall seconds spent waiting
    for the model, in the vocoder and in the metric."""
    import time
    from .datasets.spectrograms import MelSpec2Audio
    from .hparams import create_hparams
    from . import native
    from .metrics import estoi_device, stoi
    hp = create_hparams()
    fs = sampling_rate or hp.sampling_rate
    vocoder = MelSpec2Audio(hp, max_iters=max_iters, backend=vocoder_backend).to(device)
    scores = []
    was_training = net.training
    net.eval()
    t = {"model_wait_s": 0.0, "vocoder_s": 0.0, "estoi_s": 0.0, "clips": 0}
    def vocode_and_score(pending):
        """One vocoder pass over the mels of up to `group` loader batches, then ESTOI per clip.  On the device path (`metric="hip"`, default
        where the shapes allow: vocoder.hip) the vocoder is three launches per pass and the metric one block per clip - the predictions never
        leave the GPU, one (N,) score vector comes back per group; `metric="host"` scores with the numpy restatement like the reference."""
        t1 = time.perf_counter()
        same = all(m.shape[0] == pending[0][1].shape[0] for _, m in pending)
        if same:        # InverseMelScale's SGD normalises by the call's own B*L: the loader batches stay separate calls inside the one pass
            pred_dev = vocoder(torch.cat([m for _, m in pending], dim=0), rows_per_call=pending[0][1].shape[0])
        else:
            pred_dev = torch.cat([vocoder(m) for _, m in pending], dim=0)
        n = min(min(a.shape[1] for a, _ in pending), pred_dev.shape[1])      # the device branch only (all audio lengths equal there)
        on_device = metric != "host" and pred_dev.is_cuda and all(a.shape[1] == pending[0][0].shape[1] for a, _ in pending) and \
            -(-n * 10000 // fs) <= 16512
        if metric == "hip" and not on_device:
            raise RuntimeError("evaluate_net(metric='hip'): needs device predictions and clips of at most 1.65 s")
        if on_device:
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            gt_dev = torch.cat([a[:, :n] for a, _ in pending], dim=0).to(pred_dev.device, non_blocking=True).float().contiguous()
            scores.extend(float(v) for v in estoi_device(gt_dev, pred_dev[:, :n].contiguous(), fs).cpu())
            native.check_persist_timeouts()
            row = gt_dev.shape[0]
        else:
            pred = pred_dev.cpu().numpy()
            t2 = time.perf_counter()
            row = 0
            for audios, m in pending:      # per loader batch, like the reference's loop: a batch's score does not depend on which batches share its group
                gt = audios.numpy() if not audios.is_cuda else audios.cpu().numpy()
                n_b = min(gt.shape[1], pred.shape[1])
                for i in range(gt.shape[0]):
                    scores.append(stoi(gt[i, :n_b], pred[row + i, :n_b], fs, extended=True))
                row += gt.shape[0]
        t3 = time.perf_counter()
        t["vocoder_s"] += t2 - t1
        t["estoi_s"] += t3 - t2
        t["clips"] += row

    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            pending = []
            for batch, out in _evaluate_outputs(net, batches, speaker_encoder, device, group, n_inflight):
                mel = out[1]
                if pending and (len(pending) == max(1, group) or pending[0][1].shape[1:] != mel.shape[1:]):
                    t["model_wait_s"] += time.perf_counter() - t0
                    vocode_and_score(pending)
                    pending = []
                    t0 = time.perf_counter()
                pending.append((batch[1][0], mel))
            t["model_wait_s"] += time.perf_counter() - t0
            if pending:
                vocode_and_score(pending)
    finally:
        net.train(was_training)
    if timings is not None:
        timings.update(t)
    return sum(scores) / max(1, len(scores))


def reconstruction_losses(outputs, targets) -> dict:
    """`Loss.forward` (train_utils/losses.py:35-79) on torch tensors, attached to autograd: KLD of the content distribution against the
    uniform one, MSE of the pre- and (x10) post-net mels, BCE-with-logits of the stop tokens."""
    import torch.nn.functional as F
    mel_target, gate_target = targets
    mel, mel_post, stop, qy = outputs[0], outputs[1], outputs[2], outputs[5]
    return {"KLD": torch.sum(qy * torch.log(qy * qy.shape[-1] + 1e-20), dim=-1).mean(),
            "mel_loss": F.mse_loss(mel, mel_target),
            "postnet_mel_loss": 10 * F.mse_loss(mel_post, mel_target),
            "gate_loss": F.binary_cross_entropy_with_logits(stop.reshape(-1, 1), gate_target.reshape(-1, 1))}


def train_iterations(net, batches: List, n_iters: int, speaker_encoder=None, tf_ratio: float = 0.0, lr: float = 1e-4, weight_decay: float = 1e-6,
                     grad_clip: float = 1.0, fused_optimizer: bool = True, device="cuda", bf16: bool = False) -> List[dict]:
    """The model-facing half of `train.py`'s loop (train.py:102-104,150-193): `net.train()`; cycle through the collated batches
    (`tf_ratio += 0.1` every 10 epochs); forward -> 4-term loss -> `backward()` -> gradient all-reduce when a process group is up
    (one process per GPU) -> clip at `grad_clip` -> AdamW(amsgrad) on the decoder and encoder groups.  Returns the per-iteration loss
    log (python floats; the `.item()` calls are this loop's only host synchronisations, like the reference's `loss_log`).
    `bf16=True`: the training entry points run their GEMMs with bf16 operands (option "train_bf16", include/l2s.h)."""
    from .losses import Loss
    from .training import AdamWAmsgrad, GradAllReducer
    net.train()
    flat = net._train_state()
    # reduced-precision training (the reference's switch is hparams.fp16_run = apex O2, train.py:106-107,180-191): here bf16 OPERANDS in the
    # GEMMs / Conv1d stacks of encoder, prologue and post-net on the bf16 matrix cores, fp32 accumulation, fp32 master weights, fp32 loop
    net.native_model().set_option("train_bf16", 1 if bf16 else 0)
    reducer = GradAllReducer(flat.grad)          # no-op without a process group; both optimizer routes reduce (ranks must not diverge)
    # the decoder's buckets (the flat buffer holds the decoder group first) are reduced while the encoder backward still runs: the model's
    # backward calls this hook as soon as every decoder gradient is final (same overlap as bench.py --mode train)
    n_dec_buckets = reducer.buckets_covering(net._n_decoder_elems())
    net.__dict__["_on_decoder_grads"] = lambda: reducer.start(0, n_dec_buckets)
    reconstruction_criterion = Loss()
    if fused_optimizer:
        optim = AdamWAmsgrad(flat, lr=lr, weight_decay=weight_decay)
    else:
        dec, enc = net.trainable_groups()
        optim = torch.optim.AdamW([{"params": dec}, {"params": enc}], lr=lr, weight_decay=weight_decay, amsgrad=True)
    log, epoch, pos = [], 0, 0
    for _ in range(n_iters):
        if pos == len(batches):
            pos, epoch = 0, epoch + 1
            if epoch % 10 == 0:
                tf_ratio += 0.1
        (videos, vlen), (audios, alen), (melspecs, mlen, gates), face_crops = batches[pos]
        pos += 1
        emb = speaker_encoder.inference(audios.to(device)) if speaker_encoder is not None else None
        outputs = net(videos.to(device), face_crops.to(device) if face_crops is not None else None, audios.to(device), melspecs.to(device), vlen, alen,
                      mlen, tf_ratio, speaker_embedding=emb)
        losses = reconstruction_criterion(outputs, (melspecs.to(device), gates.to(device)), dict())
        loss = sum(losses.values())
        optim.zero_grad()
        loss.backward()
        reducer.start(n_dec_buckets)                 # the encoder's buckets; the decoder's are already travelling
        if fused_optimizer:
            grad_norm = optim.step(max_norm=grad_clip, grad_mul=reducer.wait())
            net.mark_weights_changed()
        else:
            mul = reducer.wait()
            if mul != 1.0:
                flat.grad.mul_(mul)                  # average over ranks before the clip (train.py:191 clips the reduced gradient)
            grad_norm = torch.nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
            optim.step()
        rec = {k: float(v.detach()) for k, v in losses.items()}
        rec.update(loss=float(loss.detach()), grad_norm=float(grad_norm), tf_ratio=tf_ratio, epoch=epoch)
        log.append(rec)
    net.__dict__["_on_decoder_grads"] = None
    return log
