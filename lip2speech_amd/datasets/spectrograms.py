"""Log-mel target transform (reference: /root/reference/datasets/spectograms.py:41-59).

The reference wraps ``torchaudio.transforms.MelSpectrogram`` (0.9.0: hann window, centre/reflect padding, power 2,
HTK mel scale, no filterbank normalisation) followed by ``log(clamp(x, 1e-5))``.  torchaudio is not available in this
environment, so the same published algorithm is restated on ``torch.stft``; its parity is therefore UNPINNED
 - it produces training targets, it is not on the measured mel-frames/s path.
"""
from __future__ import annotations

import math

import torch

from ..hparams import create_hparams


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """Triangular HTK filters, (n_freqs, n_mels), as torchaudio.functional.create_fb_matrix (norm=None)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class MelSpectrogram(torch.nn.Module):
    def __init__(self, hparams=None, n_fft=None, hop_length=None, win_length=None, n_mels=None, f_min=None, f_max=None,
                 sample_rate=None, log=True):
        super().__init__()
        hp = hparams or create_hparams()
        self.n_fft = n_fft or hp.filter_length
        self.hop = hop_length or hp.hop_length
        self.win = win_length or hp.win_length
        self.sr = sample_rate or hp.sampling_rate
        self.log = log
        n_mels = n_mels or hp.n_mel_channels
        f_min = hp.mel_fmin if f_min is None else f_min
        f_max = hp.mel_fmax if f_max is None else f_max
        self.register_buffer("window", torch.hann_window(self.win, periodic=True))
        self.register_buffer("fb", mel_filterbank(self.n_fft // 2 + 1, f_min, f_max, n_mels, self.sr))

    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        """(..., N) -> (..., n_mels, N // hop + 1)"""
        shape = waveform.shape
        x = waveform.reshape(-1, shape[-1])
        spec = torch.stft(x, self.n_fft, hop_length=self.hop, win_length=self.win, window=self.window, center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        power = spec.real ** 2 + spec.imag ** 2                                   # (B, n_freqs, L)
        mel = torch.matmul(power.transpose(1, 2), self.fb).transpose(1, 2)         # (B, n_mels, L)
        if self.log:
            mel = torch.log(torch.clamp(mel, min=1e-5))
        return mel.reshape(*shape[:-1], mel.shape[-2], mel.shape[-1])
