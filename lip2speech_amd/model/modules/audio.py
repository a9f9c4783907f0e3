"""``SpeakerEncoder`` of the boundary (reference: /root/reference/model/modules/audio.py:110-150).

Holds the ``speaker_encoder.*`` checkpoint tensors (3-layer LSTM(256) + Linear) so demo-style loaders
(``demo.py:33-43``) keep working.  Its mel front-end is torchaudio's ``MelSpectrogram`` (third-party,
"parity unpinned") and the tower is the first "next" row of SURVEY.md §8(f); until its HIP kernels
land, ``inference`` raises rather than silently computing on another path.
"""
from ... import statespec
from ._tree import ParamTree


class SpeakerEncoder(ParamTree):
    def __init__(self, state_dict=None):
        super().__init__(statespec.speaker_encoder_spec(""), key_prefix="speaker_encoder.")
        for p in self.parameters():
            p.requires_grad_(False)
        if state_dict is not None:
            self.load_state_dict(state_dict, strict=True)

    def inference(self, x):
        raise NotImplementedError("SpeakerEncoder HIP kernels are the next row of SURVEY.md §8(f); supply the "
                                  "(B,256) speaker embedding directly")

    forward = inference
