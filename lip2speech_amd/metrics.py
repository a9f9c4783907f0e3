"""ESTOI, the score `evaluate.py` reports (reference: /root/reference/evaluate.py:41-45 calls ``pystoi.stoi(clean, pred, fs,
extended=True)``).  pystoi (pinned 0.3.3, requirements.txt:6) is a third-party package that is not installed here; its published
algorithm (Jensen & Taal 2016; pystoi/stoi.py, pystoi/utils.py) is restated below on numpy/scipy - PARITY UNPINNED (SURVEY.md §8(f)
row 4).  Host-side, like the reference: the score is computed on numpy waveforms after the vocoder.

Two details of 0.3.3 that a textbook STOI does not share and that this file follows:
* framing (`remove_silent_frames`, `stft` in pystoi/utils.py) iterates ``range(0, len(x) - framelen, hop)``: the frame that would END
  exactly at the last sample is NOT taken, i.e. ceil((len - framelen) / hop) frames.  After the overlap-add of k kept frames the
  signal is (k - 1) hop + framelen long - always such a case - so the spectra have k - 1 frames, and "fewer than 30 frames -> 1e-5"
  triggers at k <= 30 kept frames;
* resampling to 10 kHz is `resample_oct`: scipy's resample_poly with pystoi's own Octave-compatible Kaiser-windowed sinc
  (`_resample_window_oct`: 60 dB rejection, cut-off 1 / (2 max(p, q)), transition a tenth of it), normalised to unit sum - not
  resample_poly's default firwin(kaiser 5.0).
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


def n_frames(n: int, framelen: int, hop: int) -> int:
    """len(range(0, n - framelen, hop)): pystoi 0.3.3's frame count (the frame ending on the last sample is dropped)."""
    return -(-(n - framelen) // hop) if n > framelen else 0


def _frames(x: np.ndarray, framelen: int, hop: int, window: np.ndarray) -> np.ndarray:
    n = n_frames(len(x), framelen, hop)
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


def resample_window_oct(p: int, q: int) -> np.ndarray:
    """pystoi/utils.py `_resample_window_oct` (a port of Octave's `resample`): Kaiser-windowed sinc, NOT normalised."""
    g = int(np.gcd(int(p), int(q)))
    p, q = int(p) // g, int(q) // g
    log10_rejection = -3.0
    stopband_cutoff_f = 1.0 / (2 * max(p, q))
    roll_off_width = stopband_cutoff_f / 10
    rejection_db = -20 * log10_rejection
    half = int(np.ceil((rejection_db - 8) / (28.714 * roll_off_width)))
    t = np.arange(-half, half + 1)
    ideal = 2 * p * stopband_cutoff_f * np.sinc(2 * stopband_cutoff_f * t)
    if 21 <= rejection_db <= 50:
        beta = 0.5842 * (rejection_db - 21) ** 0.4 + 0.07886 * (rejection_db - 21)
    elif rejection_db > 50:
        beta = 0.1102 * (rejection_db - 8.7)
    else:
        beta = 0.0
    return np.kaiser(2 * half + 1, beta) * ideal


def resample_oct(x: np.ndarray, p: int, q: int) -> np.ndarray:
    """pystoi/utils.py `resample_oct`: resample_poly with the window above normalised to unit sum."""
    h = resample_window_oct(p, q)
    return resample_poly(x, p, q, window=h / np.sum(h))


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
        x, y = resample_oct(x, FS, fs_sig), resample_oct(y, FS, fs_sig)
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
    """What `resample_oct(x, up, down)` = scipy.signal.resample_poly(x, up, down, window=h / sum(h)) computes before it filters: the FIR
    (pystoi's Octave-compatible window, unit sum, scaled by `up`) with the leading / trailing zero pads resample_poly adds, the number of
    leading outputs it drops and its output length.  Returns (h float64, up, down, n_pre_remove, n_out) with up / down reduced, or
    (None, 1, 1, 0, n_in) when there is nothing to do."""
    g = int(np.gcd(int(up), int(down)))
    up, down = int(up) // g, int(down) // g
    if up == down == 1:
        return None, 1, 1, 0, n_in
    n_out = n_in * up
    n_out = n_out // down + bool(n_out % down)
    w = resample_window_oct(up, down)
    h = w / np.sum(w) * up
    half_len = (len(h) - 1) // 2
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
