"""``Decoder`` of the boundary (reference: /root/reference/model/modules/decoder.py:274-444).

Same ``state_dict`` keys and the same two entry points:

  ``forward(encoder_outputs (B,T,1024), face_features (B,T,256), mels (B,80,S), text_lengths, output_lengths, tf_ratio)``
      -> ``[mel (B,80,S), mel_post (B,80,S), stop (B,S,1), face_features[:,0] (B,256), attention LOGITS (B,S,T),
           content_dis (B*min_T,501)]``                                       (decoder.py:320-379)
  ``inference(encoder_outputs, face_features, return_attention_map=False)``
      -> ``mel_post (B,80,300), output_lengths (B,) int64 [, attention (B,300,T)]``   (decoder.py:382-444)

Behaviour kept from the reference: lengths are ignored, ``inference`` always runs
``max_decoder_steps`` = 300 steps, and Gumbel noise is drawn even in eval mode (decoder.py:257) - here on the
device with the same ``-log(Exp(1))`` construction torch's ``F.gumbel_softmax`` uses, or supplied by the caller
through ``gumbel_noise=`` so results can be compared bit-for-bit against the reference fed the same noise.
The scheduled-sampling decision of ``forward`` (``torch.rand(1) > tf_ratio and consumed < int(tf_ratio*S)``,
decoder.py:355-357) is made here on the host, one ``torch.rand(1)`` per step exactly like the reference, and
handed to the kernel loop as a step mask.
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import native, statespec
from ...hparams import create_hparams
from ._tree import NativeBacked, ParamTree


class Decoder(ParamTree, NativeBacked):
    _key_prefix = "decoder."

    def __init__(self):
        ParamTree.__init__(self, statespec.decoder_spec(""), key_prefix="decoder.")
        self._init_native()
        self.hparams = create_hparams()
        self.n_mel_channels = self.hparams.n_mel_channels

    # -------------------------------------------------------------------------------------------------
    @staticmethod
    def draw_gumbel(rows: int, device, dtype=torch.float32) -> torch.Tensor:
        """The noise F.gumbel_softmax draws: -log(Exponential(1)) per logit."""
        return -torch.empty(rows, statespec.VOCAB, device=device, dtype=dtype).exponential_().log()

    def _prologue(self, encoder_outputs, face_features, gumbel_noise):
        nm = self.native_model()
        B, T, _ = encoder_outputs.shape
        emb = face_features[:, 0] if face_features.dim() == 3 else face_features
        rows = B * native.min_T(T)
        if gumbel_noise is None:
            gumbel_noise = self.draw_gumbel(rows, encoder_outputs.device)
        state, dis = nm.decoder_prologue(encoder_outputs, emb, gumbel_noise)
        return nm, state, dis, emb, B, T

    def _no_training(self):
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("a stand-alone Decoder is forward-only: train through Lip2Speech.forward (one autograd node over "
                                      "the HIP forward/backward of encoder + decoder) or call .eval() / torch.no_grad()")

    @staticmethod
    def sampling_mask(S: int, tf_ratio):
        """The scheduled-sampling decisions of one `forward` call, drawn exactly as decoder.py:355-357 draws them: one `torch.rand(1)` per
        step on the host; step i is fed the ground-truth previous frame iff `rand > tf_ratio and consumed < int(tf_ratio * S)`.  Returns the
        per-step byte list, or None when no step is forced (always the case at tf_ratio = 1, evaluate.py:38)."""
        mask, consumed = [], 0
        for _ in range(S):
            take = bool(torch.rand(1) > tf_ratio) and consumed < int(tf_ratio * S)
            consumed += int(take)
            mask.append(1 if take else 0)
        return mask if any(mask) else None

    def teacher_frames(self, mels: torch.Tensor) -> torch.Tensor:
        """`cat(BOS, mels)[:, :S]` channel-last (decoder.py:349): the frame fed at step i when the mask selects it."""
        B, _, S = mels.shape
        bos = self.BOS.detach().to(torch.float32).expand(B, 1, -1)
        return torch.cat([bos, mels.detach().to(torch.float32).permute(0, 2, 1)[:, :S - 1]], dim=1).contiguous()

    def forward(self, encoder_outputs, face_features, mels, text_lengths, output_lengths, tf_ratio,
                gumbel_noise: Optional[torch.Tensor] = None):
        self._no_training()
        nm, state, dis, emb, B, T = self._prologue(encoder_outputs, face_features, gumbel_noise)
        S = mels.shape[2]
        mask = self.sampling_mask(S, tf_ratio)
        teacher = self.teacher_frames(mels) if mask is not None else None
        mel, stop, attn = nm.decode_steps(state, B, T, S, teacher=teacher, teacher_mask=mask, want_attn=True, attn_logits=True)
        mel_post, mel_cf = nm.postnet(mel, want_cf=True)
        return [mel_cf, mel_post, stop.unsqueeze(2), emb, attn, dis]

    def inference(self, encoder_outputs, face_features, return_attention_map=False,
                  gumbel_noise: Optional[torch.Tensor] = None):
        with torch.no_grad():
            nm, state, _, _, B, T = self._prologue(encoder_outputs, face_features, gumbel_noise)
            S = self.hparams.max_decoder_steps
            mel, stop, attn = nm.decode_steps(state, B, T, S, want_attn=return_attention_map)
            mel_post, _ = nm.postnet(mel)
            lengths = native.output_lengths(stop)
        if return_attention_map:
            return mel_post, lengths, attn
        return mel_post, lengths
