"""``Lip2Speech`` / ``get_network`` - the drop-in boundary (reference: /root/reference/model/model.py:13-72).

``get_network(mode)`` returns an ``nn.Module`` with ``.encoder``, ``.decoder``, ``.vgg_face`` attributes,
``forward(...)`` returning the reference's list of 7 and ``inference(...)`` returning
``(mel, output_lengths[, attention])``, so train.py / evaluate.py / demo.py style callers work unchanged.
``inference`` with a supplied ``speaker_embedding`` runs as ONE native call (``l2s_inference``: encoder,
prologue, 300-step loop and post-net enqueued back to back on the current stream).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .. import native, statespec
from .modules import Decoder, FaceRecognizer, VideoExtractor
from .modules._tree import NativeBacked

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class Lip2Speech(NativeBacked):
    _key_prefix = ""

    def __init__(self):
        super().__init__()
        self._init_native()
        self.vgg_face = FaceRecognizer()
        self.encoder = VideoExtractor()
        self.decoder = Decoder()
        # one packed weight blob for the whole path; the sub-modules borrow it while they live inside this model
        self.encoder.__dict__["_native_parent"] = self
        self.decoder.__dict__["_native_parent"] = self

    def _tensors(self):
        sd = self.state_dict(keep_vars=True)
        return {k: v for k, v in sd.items() if k.startswith(("encoder.", "decoder."))}

    def _speaker(self, face_frames, speaker_embedding):
        if speaker_embedding is not None:
            return speaker_embedding
        return self.vgg_face.inference(face_frames[:, 0, :, :, :])

    def forward(self, video_frames, face_frames, audio_frames, melspecs, video_lengths, audio_lengths, melspec_lengths,
                tf_ratio, speaker_embedding=None, gumbel_noise=None):
        video_features = F.dropout(self.encoder(video_frames), 0.1, self.training)
        emb = self._speaker(face_frames, speaker_embedding)
        vis = native.build_visual(video_features, emb)
        face = emb.unsqueeze(1).expand(-1, vis.shape[1], -1)
        outputs = self.decoder(vis, face, melspecs, video_lengths, melspec_lengths, tf_ratio, gumbel_noise=gumbel_noise)
        return outputs + [video_lengths]

    def inference(self, video_frames, face_frames, speaker_embedding=None, return_attention_map=False, gumbel_noise=None):
        with torch.no_grad():
            emb = self._speaker(face_frames, speaker_embedding)
            B, _, T, _, _ = video_frames.shape
            if gumbel_noise is None:
                gumbel_noise = Decoder.draw_gumbel(B * native.min_T(T), video_frames.device)
            mel, lengths, attn = self.native_model().inference(
                video_frames, emb, gumbel_noise, S=self.decoder.hparams.max_decoder_steps, want_attn=return_attention_map)
        if return_attention_map:
            return mel, lengths, attn
        return mel, lengths


def get_network(mode):
    assert mode in ("train", "test")
    model = Lip2Speech()
    return model.train() if mode == "train" else model.eval()
