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


def spectral_de_normalize(x: torch.Tensor) -> torch.Tensor:
    """exp(x): inverse of the log compression of the targets (spectograms.py:24-39, C = 1)."""
    return torch.exp(x)


class MelSpec2Audio(torch.nn.Module):
    """Log-mel -> waveform, the vocoder `evaluate.py` scores with (reference: datasets/spectograms.py:76-95): exp, then
    torchaudio 0.9.0's ``InverseMelScale`` (non-negative least squares of ``mel = spec @ fb`` by SGD, lr 0.1 / momentum 0.9, random
    start, `max_iters` iterations with the 1e-5 / 1e-8 stopping rules) and ``GriffinLim`` (`max_iters` iterations, momentum 0.99,
    random start, power 2).  torchaudio is absent here: both published algorithms are restated on torch ops (torch.stft/istft on the
    device) - PARITY UNPINNED, and stochastic in the reference as well.  Not part of the mel-frames/s path."""

    def __init__(self, hparams=None, max_iters: int = 256, backend: str = "auto"):
        """`backend`: "hip" = the fused device kernels behind `l2s_inverse_mel` / `l2s_griffin_lim` (vocoder.hip: one wave per mel frame for
        the SGD, one block per clip with the waveform in LDS and wave-level FFTs for Griffin-Lim); "torch" = the same algorithms as ~5 000
        torch launches per call (the restatement the kernels are tested against); "auto" = "hip" for device tensors of a supported shape
        (n_fft = win = 1024, hop 256, at most 121 frames per clip), else "torch".  Both draw the same random start iterates in the same order."""
        super().__init__()
        hp = hparams or create_hparams()
        self.n_fft, self.hop, self.win, self.sr = hp.filter_length, hp.hop_length, hp.win_length, hp.sampling_rate
        self.max_iters = max_iters
        self.backend = backend
        self.register_buffer("window", torch.hann_window(self.win, periodic=True))
        self.register_buffer("fb", mel_filterbank(self.n_fft // 2 + 1, hp.mel_fmin, hp.mel_fmax, hp.n_mel_channels, self.sr))
        self.fb_nnz = int((self.fb != 0).sum())

    def _use_hip(self, t: torch.Tensor, L: int) -> bool:
        ok = t.is_cuda and self.n_fft == 1024 and self.win == 1024 and self.hop == 256 and 5 <= L <= 121 and self.fb_nnz <= 2048
        if self.backend == "hip" and not ok:
            raise RuntimeError("MelSpec2Audio(backend='hip'): needs device tensors, n_fft = win = 1024, hop 256 and 5..121 frames per clip")
        return ok and self.backend in ("auto", "hip")

    # -- torchaudio.transforms.InverseMelScale.forward (0.9.0)
    def inverse_mel(self, mel: torch.Tensor, generator=None, rows_per_call: int = None) -> torch.Tensor:
        """mel (B, n_mels, L) power-mel -> power spectrogram (B, n_freqs, L) >= 0.  The reference's loop reads its loss on the host every
        iteration (`new_loss.item()`) for the two stopping rules; here the rules are evaluated on the device and freeze the iterate from the
        stopping iteration on, so the 256 iterations are enqueued without a single host synchronisation and end at the same iterate.
        `rows_per_call`: the rows are G independent calls of that many rows each (several loader batches vocoded in one pass): the loss mean,
        the 1/(B*L) gradient scale and the stopping rules are per call, exactly as G separate calls would have them."""
        N, _, L = mel.shape
        R = rows_per_call or N
        assert N % R == 0, "rows_per_call must divide the batch"
        G = N // R
        target = mel.transpose(1, 2).reshape(G, R * L, -1)                        # (G, R*L, n_mels)
        spec = torch.rand(G, R * L, self.fb.shape[0], device=mel.device, dtype=mel.dtype, generator=generator)
        if self._use_hip(mel, L):
            from .. import native
            return native.inverse_mel(mel, self.fb, self.fb_nnz, spec.reshape(N * L, -1), self.max_iters, rows_per_call=R)
        vel = torch.zeros_like(spec)
        loss = torch.full((G, 1, 1), float("inf"), device=mel.device, dtype=mel.dtype)
        active = torch.ones((G, 1, 1), device=mel.device, dtype=torch.bool)
        fbt = self.fb.t().contiguous()
        for _ in range(self.max_iters):
            diff = target - spec @ self.fb
            new_loss = diff.pow(2).sum(dim=-1).mean(dim=-1).view(G, 1, 1)
            grad = (-2.0 / (R * L)) * (diff @ fbt)                                # d/dspec of the call's mean over (B, L) of the per-frame squared error
            new_vel = 0.9 * vel + grad                                            # torch.optim.SGD(lr=0.1, momentum=0.9)
            new_spec = (spec - 0.1 * new_vel).clamp_(min=0)
            vel = torch.where(active, new_vel, vel)                               # the update of the iteration that meets a rule is still applied
            spec = torch.where(active, new_spec, spec)
            active = active & ~((new_loss < 1e-5) | ((loss - new_loss).abs() < 1e-8))
            loss = new_loss
        return spec.reshape(N, L, -1).transpose(1, 2)

    # -- torchaudio.functional.griffinlim (0.9.0): power 2, momentum 0.99, rand_init
    def griffin_lim(self, power_spec: torch.Tensor, generator=None) -> torch.Tensor:
        mag = power_spec.clamp(min=0).sqrt()                                      # (B, n_freqs, L)
        B, _, L = mag.shape
        length = self.hop * (L - 1)
        ang = torch.view_as_complex(torch.rand(*mag.shape, 2, device=mag.device, dtype=mag.dtype, generator=generator))
        if self._use_hip(power_spec, L):
            from .. import native
            return native.griffin_lim(power_spec, ang, self.max_iters, self.n_fft, self.hop, 0.99)
        momentum = 0.99 / (1 + 0.99)
        prev = torch.zeros_like(ang)
        stft = lambda x: torch.stft(x, self.n_fft, self.hop, self.win, self.window, center=True, pad_mode="reflect",      # noqa: E731
                                    normalized=False, onesided=True, return_complex=True)
        istft = lambda z: torch.istft(z, self.n_fft, self.hop, self.win, self.window, length=length)                      # noqa: E731
        for _ in range(self.max_iters):
            rebuilt = stft(istft(mag * ang))
            ang = rebuilt - prev * momentum
            ang = ang / (ang.abs() + 1e-16)
            prev = rebuilt
        return istft(mag * ang)

    def forward(self, melspec: torch.Tensor, generator=None, rows_per_call: int = None) -> torch.Tensor:
        """(B, n_mels, L) log-mel -> (B, hop*(L-1)) waveform.  `rows_per_call`: see `inverse_mel` (Griffin-Lim is per clip anyway)."""
        mel = spectral_de_normalize(melspec.to(torch.float32))
        return self.griffin_lim(self.inverse_mel(mel, generator, rows_per_call), generator)
