"""Data boundary of the path (reference: /root/reference/datasets/__init__.py:7-89, datasets/spectograms.py:41-59).

Host-side only: these functions define the tensor layouts that cross into the model -
``(B,3,T,96,96)`` frames zero-padded to the batch maximum, mel targets padded with ln(1e-5), gate = 1 from the
last real frame on.  The model ignores the returned lengths, so the padding IS part of the
semantics and is reproduced exactly.
"""
from __future__ import annotations

import torch

from .spectrograms import MelSpec2Audio, MelSpectrogram  # noqa: F401
from .lrw import LRW  # noqa: F401
from .augmentation import FaceAugmentation  # noqa: F401
from .unported import GRID, AVSpeech, WILD  # noqa: F401
from .device import PackedFrames, device_collate_fn_pad  # noqa: F401

MEL_PAD = -11.5129      # ln(1e-5), the floor of the log-mel transform


def _pad_audio_mels(speeches, melspecs, mel_pad: float = MEL_PAD):
    """((audio (n,a_max), lengths), (mels (n,80,m_max) padded with `mel_pad`, lengths, gate = 1 from the last real frame on)).  The LRW /
    top-level collates pad mel targets with ln(1e-5) (datasets/__init__.py:20); the per-corpus `av_speech_collate_fn_pad` pads with 0."""
    n = len(speeches)
    a_max = max(s.shape[1] for s in speeches)
    m_max = max(m.shape[1] for m in melspecs)
    audio = torch.zeros(n, a_max)
    mels = torch.full((n, melspecs[0].shape[0], m_max), float(mel_pad))
    gate = torch.zeros(n, m_max)
    a_len, m_len = [], []
    for i, (speech, mel) in enumerate(zip(speeches, melspecs)):
        audio[i, :speech.shape[-1]] = speech.reshape(-1)
        mels[i, :, :mel.shape[-1]] = mel
        gate[i, mel.shape[-1] - 1:] = 1.0
        a_len.append(speech.shape[-1]); m_len.append(mel.shape[-1])
    return (audio, torch.tensor(a_len)), (mels, torch.tensor(m_len), gate)


def _collate(batch, with_paths: bool, mel_pad: float = MEL_PAD):
    if with_paths:
        mouths, speeches, melspecs, faces, paths = zip(*batch)
    else:
        mouths, speeches, melspecs, faces = zip(*batch)
        paths = None
    n = len(mouths)
    t_max = max(m.shape[0] for m in mouths)
    video = torch.zeros(n, t_max, *mouths[0].shape[1:])
    for i, mouth in enumerate(mouths):
        video[i, :mouth.shape[0]] = mouth
    out = (((video.permute(0, 2, 1, 3, 4), torch.tensor([m.shape[0] for m in mouths])),) + _pad_audio_mels(speeches, melspecs, mel_pad) +
           (torch.stack(list(faces), dim=0),))
    return out + (paths,) if with_paths else out


def train_collate_fn_pad(batch):
    """[(mouth (T,3,H,W), speech (1,N), melspec (80,M), face_crop (2,3,160,160))] -> the reference's 4-tuple."""
    return _collate(batch, with_paths=False)


def test_collate_fn_pad(batch):
    """Same with the per-item file paths appended (demo.py:60)."""
    return _collate(batch, with_paths=True)


test_collate_fn_pad.__test__ = False      # not a pytest test despite the reference's name


def av_speech_collate_fn_pad(batch):
    """The per-corpus collate of `datasets.{grid,avspeech,wild}` (reference: datasets/grid/dataset.py:28-68, avspeech/dataset.py:53-93,
    wild/dataset.py:35): the same 4-tuple, but padded mel-target frames are ZERO there, not ln(1e-5) - a different loss on padded batches."""
    return _collate(batch, with_paths=False, mel_pad=0.0)


def av_speech_collate_fn_trim(batch):
    """datasets/avspeech/dataset.py:30-50 (marked by the reference itself as not working with its current code; kept for the import
    surface): trims every clip and waveform to the batch minimum instead of padding; returns
    ((frames (n,T_min,3,H,W), [T_min]*n), (speech (n,1,N_min), [N_min]*n), face_crops)."""
    mouths, speeches, _melspecs, faces = zip(*batch)
    n = len(mouths)
    t_min = min(m.shape[0] for m in mouths)
    a_min = min(s.shape[1] for s in speeches)
    frames = torch.stack([m[:t_min] for m in mouths], dim=0)
    speech = torch.stack([s[:, :a_min] for s in speeches], dim=0)
    return (frames, [t_min] * n), (speech, [a_min] * n), torch.stack(list(faces), dim=0)
