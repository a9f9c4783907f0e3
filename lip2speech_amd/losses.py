"""`Loss` of the training step (reference: /root/reference/train_utils/losses.py:13-79) over the HIP kernel `l2s_loss`.

Same call signature and result as the reference: ``Loss()(model_output, (mel_target, gate_target), losses=None)`` returns the dict
``{'KLD', 'mel_loss', 'postnet_mel_loss', 'gate_loss'}`` (train.py:172-176 sums its values).  The four terms and their gradients come
from ONE launch chain of `l2s_loss` (two-stage deterministic reductions); each term is a separate autograd output, so
``sum(losses.values()).backward()`` - or any weighting of the terms - flows into ``Lip2Speech.forward``'s HIP backward.
There is no CPU path: host tensors raise.
"""
from __future__ import annotations

import torch
from torch import nn

from . import training

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")      # the reference module exports this name (losses.py:10)


class _LossTerms(torch.autograd.Function):
    """(mel, mel_post, stop, content_dis) -> (mel_loss, postnet_mel_loss, gate_loss, KLD); every input feeds exactly one term."""

    @staticmethod
    def forward(ctx, mel, mel_post, stop, content_dis, mel_target, gate_target):
        out, g = training.loss_terms(mel, mel_post, stop, content_dis, mel_target, gate_target, want_grads=True)
        ctx.save_for_backward(g["mel"], g["mel_post"], g["stop"], g["content_dis"])
        ctx.shapes = (mel.shape, mel_post.shape, stop.shape, content_dis.shape)
        return out[0].clone(), out[1].clone(), out[2].clone(), out[3].clone()

    @staticmethod
    def backward(ctx, d_mel, d_post, d_gate, d_kld):
        g_mel, g_post, g_stop, g_dis = ctx.saved_tensors
        s = ctx.shapes
        return ((g_mel * d_mel).reshape(s[0]), (g_post * d_post).reshape(s[1]), (g_stop * d_gate).reshape(s[2]),
                (g_dis * d_kld).reshape(s[3]), None, None)


class Loss(nn.Module):
    def __init__(self):
        super().__init__()
        self.attention_mask = self.LRW_attention_mask()

    def LRW_attention_mask(self):
        """(1,77) int64: the diagonal alignment target `int(i / 77 * 29)` the reference builds and (in its current form) never uses
        (losses.py:22-33; the attention loss at :64-66 is commented out)."""
        seq_len, inp_len = 77, 29
        return torch.tensor([[int((i / seq_len) * inp_len) for i in range(seq_len)]], dtype=torch.long)

    def forward(self, model_output, targets, losses=None):
        if losses is None:
            losses = dict()
        mel_target, gate_target = targets[0], targets[1]
        mel, mel_post, gate_out, qy = model_output[0], model_output[1], model_output[2], model_output[5]
        if not mel.is_cuda:
            raise RuntimeError("train_utils.losses.Loss runs on the GPU (l2s_loss): move the model outputs and targets to cuda (no CPU fallback)")
        m, p, g, k = _LossTerms.apply(mel, mel_post, gate_out, qy, mel_target.to(mel.device), gate_target.to(mel.device))
        losses["KLD"] = k
        losses["mel_loss"] = m
        losses["postnet_mel_loss"] = p
        losses["gate_loss"] = g
        return losses
