"""ESTOI, the score `evaluate.py` reports (reference: /root/reference/evaluate.py:41-45 calls ``pystoi.stoi(clean, pred, fs,
extended=True)``).  pystoi (0.3.3) is a third-party package that is not installed here; its published algorithm (Jensen & Taal 2016;
pystoi/stoi.py, pystoi/utils.py) is restated below on numpy/scipy - PARITY UNPINNED.  Host-side, like the
reference: the score is computed on numpy waveforms after the vocoder.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import resample_poly

FS = 10000          # sample rate of the intelligibility model
N_FRAME = 256       # window
NFFT = 512
NUMBAND = 15
MINFREQ = 150
N = 30              # frames per intermediate intelligibility segment (384 ms)
DYN_RANGE = 40      # dB, silent-frame removal
EPS = np.finfo("float").eps


def thirdoct(fs: int, nfft: int, num_bands: int, min_freq: float):
    """1/3-octave band matrix (num_bands, nfft//2+1) and centre frequencies."""
    f = np.linspace(0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(num_bands, dtype=float)
    cf = 2.0 ** (k / 3.0) * min_freq
    lo = min_freq * 2.0 ** ((2 * k - 1) / 6)
    hi = min_freq * 2.0 ** ((2 * k + 1) / 6)
    obm = np.zeros((num_bands, len(f)))
    for i in range(num_bands):
        lo_i = int(np.argmin((f - lo[i]) ** 2))
        hi_i = int(np.argmin((f - hi[i]) ** 2))
        obm[i, lo_i:hi_i] = 1
    return obm, cf


def _frames(x: np.ndarray, framelen: int, hop: int, window: np.ndarray) -> np.ndarray:
    n = 1 + (len(x) - framelen) // hop if len(x) >= framelen else 0
    idx = np.arange(framelen)[None, :] + hop * np.arange(n)[:, None]
    return x[idx] * window[None, :]


def remove_silent_frames(x: np.ndarray, y: np.ndarray, dyn_range: float, framelen: int, hop: int):
    """Drop the frames of the clean signal x (and the same frames of y) whose energy is more than dyn_range dB below the loudest
    frame, then overlap-add the rest."""
    w = np.hanning(framelen + 2)[1:-1]
    xf, yf = _frames(x, framelen, hop, w), _frames(y, framelen, hop, w)
    if len(xf) == 0:
        return x, y
    energies = 20 * np.log10(np.linalg.norm(xf, axis=1) + EPS)
    keep = (np.max(energies) - dyn_range - energies) < 0
    xf, yf = xf[keep], yf[keep]
    n_out = (len(xf) - 1) * hop + framelen if len(xf) else 0
    xo, yo = np.zeros(n_out), np.zeros(n_out)
    for i in range(len(xf)):
        xo[i * hop: i * hop + framelen] += xf[i]
        yo[i * hop: i * hop + framelen] += yf[i]
    return xo, yo


def _stft(x: np.ndarray, win_size: int, fft_size: int, overlap: int = 2) -> np.ndarray:
    hop = win_size // overlap
    w = np.hanning(win_size + 2)[1:-1]
    return np.fft.rfft(_frames(x, win_size, hop, w), n=fft_size)           # (frames, fft_size//2+1)


def _row_col_normalize(x: np.ndarray) -> np.ndarray:
    """x: (segments, bands, N): zero-mean unit-norm rows, then columns."""
    x = x - x.mean(axis=-1, keepdims=True)
    x = x / (np.linalg.norm(x, axis=-1, keepdims=True) + EPS)
    x = x - x.mean(axis=1, keepdims=True)
    x = x / (np.linalg.norm(x, axis=1, keepdims=True) + EPS)
    return x


def stoi(x, y, fs_sig: int, extended: bool = False) -> float:
    """(E)STOI of the processed signal y against the clean signal x (both 1-D, same length, sample rate fs_sig)."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if x.shape != y.shape:
        raise Exception("x and y should have the same length, found {} and {}".format(x.shape, y.shape))
    if fs_sig != FS:
        g = np.gcd(FS, fs_sig)
        x, y = resample_poly(x, FS // g, fs_sig // g), resample_poly(y, FS // g, fs_sig // g)
    x, y = remove_silent_frames(x, y, DYN_RANGE, N_FRAME, N_FRAME // 2)
    xs, ys = _stft(x, N_FRAME, NFFT).T, _stft(y, N_FRAME, NFFT).T          # (freq, frames)
    if xs.shape[-1] < N:
        return 1e-5                                                         # pystoi warns and returns 1e-5: not enough frames
    obm, _ = thirdoct(FS, NFFT, NUMBAND, MINFREQ)
    x_tob = np.sqrt(obm @ np.abs(xs) ** 2)                                   # (bands, frames)
    y_tob = np.sqrt(obm @ np.abs(ys) ** 2)
    n_seg = x_tob.shape[1] - N + 1
    x_seg = np.stack([x_tob[:, m: m + N] for m in range(n_seg)])            # (segments, bands, N)
    y_seg = np.stack([y_tob[:, m: m + N] for m in range(n_seg)])
    if extended:
        xn, yn = _row_col_normalize(x_seg), _row_col_normalize(y_seg)
        return float(np.sum(xn * yn / N) / xn.shape[0])
    # classic STOI: clip the processed segments at -15 dB signal-to-distortion, then average the band correlations
    beta = -15.0
    norm = np.linalg.norm(x_seg, axis=2, keepdims=True) / (np.linalg.norm(y_seg, axis=2, keepdims=True) + EPS)
    y_prim = np.minimum(y_seg * norm, x_seg * (1 + 10 ** (-beta / 20)))
    xc = x_seg - x_seg.mean(axis=2, keepdims=True)
    yc = y_prim - y_prim.mean(axis=2, keepdims=True)
    xc = xc / (np.linalg.norm(xc, axis=2, keepdims=True) + EPS)
    yc = yc / (np.linalg.norm(yc, axis=2, keepdims=True) + EPS)
    return float(np.sum(xc * yc) / (xc.shape[0] * xc.shape[1]))


# ---------------------------------------------------------------------------------------------------------------- on the device (l2s_estoi)
def resample_poly_plan(n_in: int, up: int, down: int):
    """What scipy.signal.resample_poly(x, up, down) (default Kaiser-5 window, zero-padded ends) computes before it filters: the FIR it
    designs - scaled by `up`, with its leading / trailing zero pads - the number of leading outputs it drops and its output length.
    Returns (h float64, up, down, n_pre_remove, n_out) with up / down reduced, or (None, 1, 1, 0, n_in) when there is nothing to do."""
    from scipy.signal import firwin
    g = int(np.gcd(int(up), int(down)))
    up, down = int(up) // g, int(down) // g
    if up == down == 1:
        return None, 1, 1, 0, n_in
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
    n_pre_pad = down - half_len % down
    n_post_pad = 0
    n_pre_remove = (half_len + n_pre_pad) // down
    out_len = lambda len_h: (((n_in - 1) * up + len_h) - 1) // down + 1      # noqa: E731  (scipy.signal._upfirdn._output_len)
    while out_len(len(h) + n_pre_pad + n_post_pad) < n_out + n_pre_remove:
        n_post_pad += 1
    h = np.concatenate((np.zeros(n_pre_pad), h, np.zeros(n_post_pad)))
    return h, up, down, n_pre_remove, n_out


def band_edges():
    """First / last+1 bin of the 15 one-third octave bands (rows of `thirdoct`'s matrix), as l2s_estoi takes them: lo[0..14] + hi[0..14]."""
    obm, _ = thirdoct(FS, NFFT, NUMBAND, MINFREQ)
    lo, hi = [], []
    for row in obm:
        nz = np.flatnonzero(row)
        lo.append(int(nz[0]) if len(nz) else 0)
        hi.append(int(nz[-1]) + 1 if len(nz) else 0)
        assert len(nz) == 0 or np.all(row[lo[-1]:hi[-1]] == 1), "bands are contiguous runs of ones"
    return lo + hi


_plans = {}


def estoi_device(clean, pred, fs_sig: int):
    """ESTOI of every row of `pred` against the same row of `clean` (device tensors (N, n_samples)) by the HIP kernel behind `l2s_estoi`:
    the algorithm of `stoi(x, y, fs_sig, extended=True)` above, one block per clip.  Returns a device tensor (N,)."""
    import torch
    from . import native
    n = int(clean.shape[1])
    key = (n, int(fs_sig), str(clean.device))
    if key not in _plans:
        h, up, down, n_pre, n_out = resample_poly_plan(n, FS, fs_sig)
        fir = None if h is None else torch.from_numpy(h.astype(np.float32)).to(clean.device)
        _plans[key] = (fir, up, down, n_pre, n_out, band_edges())
    fir, up, down, n_pre, n_out, bands = _plans[key]
    return native.estoi(clean, pred, fir, up, down, n_pre, n_out, bands)
