"""Training-side pieces of the data-parallel step (reference: /root/reference/train.py:102-104,172-193 and
train_utils/losses.py:35-79), built so far: the 4-term loss (with its gradients), global-norm clipping + AdamW(amsgrad)
as one fused HIP update over a flat parameter buffer, and the bucketed gradient all-reduce over RCCL (SURVEY.md §8(e):
one all-reduce of 38 436 836 fp32 gradients per step, in ~25 MB buckets so the first buckets travel over xGMI while later
ones are still being produced).  The backward kernels of the model itself are the next row; these pieces
are independent of how the gradient buffer gets filled and are tested on their own.
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
        off = 0
        self.offsets: Dict[int, int] = {}
        for p in self.params:
            n = p.numel()
            self.data[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.data[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            self.offsets[id(p)] = off
            off += n


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


class GradAllReducer:
    """Sum all-reduce of a flat gradient buffer in fixed-size buckets (default 25 MB), issued asynchronously in order;
    `wait()` blocks on all of them.  The division by world size is NOT done here - AdamWAmsgrad.step(grad_mul=1/world)
    folds it into the update, and clipping uses the norm of the averaged gradient (train.py:191 clips after the reduce)."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 25 * 1024 * 1024):
        self.buf = flat_grad
        per = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets = [flat_grad[i:i + per] for i in range(0, flat_grad.numel(), per)]
        self._work = []

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
