"""``SpeakerEncoder`` of the boundary (reference: /root/reference/model/modules/audio.py:110-150).

Holds the ``speaker_encoder.*`` checkpoint tensors (3-layer LSTM(256) + Linear) so demo-style loaders
(``demo.py:33-43``: ``SpeakerEncoder(state_dict=...)`` then ``.inference(audios)``) keep working, and runs
``inference`` on the device through ``l2s_speaker_encoder_fwd``: 40-band mel front-end (DFT and filterbank as fp32
MFMA GEMMs), three LSTM layers on the batch-row LSTM kernel, Linear + ReLU + L2 normalisation.
The mel front-end restates torchaudio 0.9's published algorithm (third-party, absent here): parity UNPINNED for
that piece; the LSTM/Linear tail is checked against ``torch.nn.LSTM`` in the tests.
"""
import torch

from ... import statespec
from ._tree import NativeBacked, ParamTree


class SpeakerEncoder(ParamTree, NativeBacked):
    _key_prefix = "speaker_encoder."

    def __init__(self, state_dict=None):
        ParamTree.__init__(self, statespec.speaker_encoder_spec(""), key_prefix="speaker_encoder.")
        self._init_native()
        for p in self.parameters():
            p.requires_grad_(False)
        if state_dict is not None:
            self.load_state_dict(state_dict, strict=True)

    def inference(self, x: torch.Tensor) -> torch.Tensor:
        """audio (B, n_samples) at 16 kHz -> (B,256) non-negative unit-norm embedding."""
        if self.training:
            self.eval()
        with torch.no_grad():
            return self.native_model().speaker_encoder_fwd(x)

    def forward(self, utterances, hidden_init=None):
        raise NotImplementedError("only SpeakerEncoder.inference (the frozen, eval-mode use in demo.py:84) is built on the HIP path")
