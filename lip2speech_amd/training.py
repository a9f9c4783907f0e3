"""Training-side pieces of the data-parallel step (reference: /root/reference/train.py:102-104,172-193 and
train_utils/losses.py:35-79): the 4-term loss (with its gradients), the forward+backward drivers of the decoder and of the whole
model over the HIP kernels, global-norm clipping + AdamW(amsgrad) as one fused HIP update over a flat parameter buffer, and the
bucketed gradient all-reduce over RCCL (one all-reduce of 38 436 836 fp32 gradients per step, in ~25 MB buckets).
DESIGN.md §9 describes the path; tests/test_grad_goldens.py pins it to the reference's own backward.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import native


class FlatBuffer:
    """Parameters re-homed into ONE contiguous fp32 device buffer (group order as train.py:102-104: decoder, then
    encoder), each nn.Parameter becoming a view; a second buffer of the same shape holds the gradients."""

    def __init__(self, groups: Sequence[Iterable[torch.nn.Parameter]]):
        self.params: List[torch.nn.Parameter] = [p for g in groups for p in g]
        assert self.params, "no parameters"
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.data = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        # `grad._version` at the moment a fused zero_grad() zeroed the buffer, else -1.  While it still equals `grad._version` nothing has written
        # the gradient buffer through torch since (every in-place op on the buffer or on any p.grad view bumps the shared version counter), so
        # the next backward() may overwrite instead of clone + add; the backward itself (native writes, invisible to the counter) resets it.
        self.cleared_version = -1
        off = 0
        self.offsets: Dict[int, int] = {}
        for p in self.params:
            n = p.numel()
            self.data[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            self.offsets[id(p)] = off
            off += n


    def mark_cleared(self) -> None:
        self.cleared_version = self.grad._version

    def is_cleared(self) -> bool:
        """True iff the gradient buffer is still the zeros a fused zero_grad() left (see `cleared_version`)."""
        return self.cleared_version == self.grad._version

    def mark_written(self) -> None:
        self.cleared_version = -1


class AdamWAmsgrad:
    """torch.optim.AdamW(params, lr, weight_decay, amsgrad=True) + clip_grad_norm_(max_norm) as two launches:
    l2s_grad_norm (deterministic two-stage reduction) and l2s_adamw_amsgrad_step (update fused with clip and 1/world)."""

    def __init__(self, flat: FlatBuffer, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        z = lambda: torch.zeros_like(flat.data)      # noqa: E731
        self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq = z(), z(), z()
        self.step_count = 0
        self._scratch = torch.empty(int(native.lib().l2s_train_scratch_bytes()), dtype=torch.uint8, device=flat.data.device)
        self._norm = torch.zeros(1, dtype=torch.float32, device=flat.data.device)

    def zero_grad(self):
        self.flat.grad.zero_()
        self.flat.mark_cleared()          # the next backward() overwrites: no clone + add of a zero buffer (154 MB each) for "accumulation"

    def step(self, max_norm: Optional[float] = 1.0, grad_mul: float = 1.0) -> torch.Tensor:
        """Returns the (unclipped, averaged) total gradient norm as a device scalar - no host synchronisation."""
        L, f = native.lib(), self.flat
        self.step_count += 1
        s = torch.cuda.current_stream().cuda_stream
        native.check(L.l2s_grad_norm(f.grad.data_ptr(), f.numel, self._scratch.data_ptr(), self._norm.data_ptr(), s))
        native.check(L.l2s_adamw_amsgrad_step(f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                              self.max_exp_avg_sq.data_ptr(), f.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                              self.step_count, self._norm.data_ptr() if max_norm else None, grad_mul,
                                              float(max_norm or 0.0), s))
        return self._norm * grad_mul


def loss_terms(mel, mel_post, stop, content_dis, mel_target, gate_target, want_grads: bool = True):
    """The reference's `Loss.forward` (losses.py:69-77) on the device: returns ({'mel_loss','postnet_mel_loss','gate_loss','KLD',
    'loss'} as a (5,) tensor, gradients dict).  mel/mel_post/mel_target (B,80,S); stop (B,S[,1]); gate (B,S); content_dis (R,501)."""
    L = native.lib()
    f32 = lambda t: t.detach().to(torch.float32).contiguous()      # noqa: E731
    mel, mel_post, mel_target, content_dis = f32(mel), f32(mel_post), f32(mel_target), f32(content_dis)
    stop = f32(stop).reshape(mel.shape[0], -1)
    gate_target = f32(gate_target)
    B, _, S = mel.shape
    R = content_dis.shape[0]
    out = torch.empty(5, dtype=torch.float32, device=mel.device)
    scratch = torch.empty(int(L.l2s_train_scratch_bytes()), dtype=torch.uint8, device=mel.device)
    grads = {}
    if want_grads:
        grads = {"mel": torch.empty_like(mel), "mel_post": torch.empty_like(mel_post), "stop": torch.empty_like(stop),
                 "content_dis": torch.empty_like(content_dis)}
    gp = lambda k: grads[k].data_ptr() if want_grads else None      # noqa: E731
    native.check(L.l2s_loss(mel.data_ptr(), mel_post.data_ptr(), mel_target.data_ptr(), stop.data_ptr(), gate_target.data_ptr(),
                            content_dis.data_ptr(), B, S, R, out.data_ptr(), gp("mel"), gp("mel_post"), gp("stop"), gp("content_dis"),
                            scratch.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out, grads


def draw_dropout(B: int, T: int, S: int, device, generator=None) -> dict:
    """The train-mode dropout multipliers of one `Lip2Speech.forward` (0 or 1/(1-p)) as explicit tensors, in the order the reference
    consumes its RNG: features p 0.1 (model.py:26), per step prenet p 0.2 / attention logits p 0.1 / LSTM inter-layer
    p 0.1 (decoder.py:308,363,312), post-net p 0.5 x5 (:152,154).  torch's own Philox streams are not reproduced bit for bit (that
    would need the reference's exact draw shapes on its device); the masks are inputs, like the Gumbel noise, so a run is reproducible
    and comparable against the oracle."""
    def mk(shape, p):
        return (torch.rand(shape, device=device, generator=generator) >= p).to(torch.float32) / (1.0 - p)
    d = {"feat": mk((B, T, 768), 0.1), "prenet": mk((S, B, 256), 0.2), "attn": mk((S, B, T), 0.1), "rnn": mk((S, B, 512), 0.1)}
    d["post"] = [mk((B, 512 if l < 4 else 80, S), 0.5) for l in range(5)]
    return d


def decoder_forward_backward(nm: "native.NativeModel", vis, emb, gumbel, mel_target, gate_target, teacher_mask=None, bos=None, drop=None):
    """Forward + backward of the decoder half of `Lip2Speech.forward` + `Loss.forward` (reference: model.py:34-41 -> decoder.py:320-379,
    losses.py:69-77, train.py:172-184) with eval-mode statistics (running BN stats, no dropout): prologue -> S-step loop -> post-net ->
    4-term loss, then back through the post-net, the loop (BPTT) and the prologue.  Parameter gradients land in the slots bound with
    `nm.train_bind`; returns the outputs, the loss terms and the gradient wrt the visual features `vis` (B,T,1024).

    teacher_mask (S,) bool marks the steps whose input frame is the ground-truth previous frame (scheduled sampling made explicit,
    decoder.py:355-359); `bos` is the device BOS parameter (needed only with a mask)."""
    B, T, _ = vis.shape
    S = mel_target.shape[2]
    drop = drop or {}
    pdrop = native.postnet_drop_pack(drop["post"]) if drop.get("post") is not None else None
    if drop.get("feat") is not None:                      # F.dropout on the encoder features (model.py:26); the embedding columns pass
        vis = vis.clone()
        vis[:, :, :768] *= drop["feat"]
    state, dis, ptape = nm.train_prologue_fwd(vis, emb, gumbel)
    teacher = None
    if teacher_mask is not None and bool(torch.as_tensor(teacher_mask).any()):
        assert bos is not None, "teacher forcing needs the BOS parameter"
        teacher = torch.cat([bos.reshape(1, 1, 80).expand(B, 1, 80), mel_target.permute(0, 2, 1)[:, :S - 1]], dim=1).contiguous()
        teacher_mask = torch.as_tensor(teacher_mask).cpu().numpy()
    else:
        teacher_mask = None
    (mel, stop, logits), ctx = nm.train_steps_fwd(state, B, T, S, teacher, teacher_mask, drop=drop)
    mel_post, post_tape = nm.train_postnet_fwd(mel, pdrop)
    mel_cf = mel.permute(0, 2, 1).contiguous()
    loss, g = loss_terms(mel_cf, mel_post, stop, dis, mel_target, gate_target)
    wbuf = nm.train_pack_weights(vis.device)
    dmel = nm.train_postnet_bwd(mel, g["mel_post"], post_tape, pdrop)
    dmel += g["mel"].permute(0, 2, 1)
    sg = nm.train_steps_bwd(ctx, dmel, g["stop"], wbuf=wbuf)
    dvis = nm.train_prologue_bwd(vis, emb, state, ptape, sg, dcontent_dis=g["content_dis"], wbuf=wbuf)
    if drop.get("feat") is not None:
        dvis[:, :, :768] *= drop["feat"]
    return {"loss": loss, "mel": mel_cf, "mel_post": mel_post, "stop": stop, "attn_logits": logits, "content_dis": dis, "dvis": dvis}


def model_forward_backward(nm: "native.NativeModel", video, emb, gumbel, mel_target, gate_target, teacher_mask=None, bos=None, drop=None,
                           on_decoder_grads=None):
    """`Lip2Speech.forward` + `Loss` + `backward()` (model.py:20-41, train.py:167-184) with the speaker embedding supplied and eval-mode
    statistics: encoder forward with a tape, `decoder_forward_backward`, then the encoder backward fed by the visual-feature gradient.
    All encoder and decoder parameter gradients land in the bound slots."""
    vis, _, etape = nm.train_encoder_fwd(video, emb)
    out = decoder_forward_backward(nm, vis, emb, gumbel, mel_target, gate_target, teacher_mask=teacher_mask, bos=bos, drop=drop)
    if on_decoder_grads is not None:
        on_decoder_grads()            # every decoder gradient is final: their all-reduce can travel under the encoder backward
    nm.train_encoder_bwd(video, out["dvis"], etape)
    return out


class GradAllReducer:
    """Sum all-reduce of a flat gradient buffer in fixed-size buckets (default 25 MB), issued asynchronously in order;
    `wait()` blocks on all of them.  The division by world size is NOT done here - AdamWAmsgrad.step(grad_mul=1/world)
    folds it into the update, and clipping uses the norm of the averaged gradient (train.py:191 clips after the reduce)."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 25 * 1024 * 1024):
        self.buf = flat_grad
        per = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets = [flat_grad[i:i + per] for i in range(0, flat_grad.numel(), per)]
        self._work = []

    def buckets_covering(self, numel: int) -> int:
        """Number of leading buckets that lie entirely inside the first `numel` elements of the flat buffer."""
        n, count = 0, 0
        for b in self.buckets:
            if n + b.numel() > numel:
                break
            n += b.numel()
            count += 1
        return count

    def start(self, first: int = 0, last: Optional[int] = None):
        """Launch the all-reduce of buckets [first, last) - call as soon as those gradients are final."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for b in self.buckets[first:last]:
            self._work.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self) -> float:
        for w in self._work:
            w.wait()
        self._work = []
        return 1.0 / (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1)
