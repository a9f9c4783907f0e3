"""Vocoder + ESTOI, the tail of `evaluate.py`.  Both are restatements of third-party algorithms whose
packages (torchaudio 0.9.0, pystoi 0.3.3) are absent here - parity unpinned; these tests pin their defining properties."""
import numpy as np
import pytest
import torch

from lip2speech_amd import metrics
from lip2speech_amd.datasets.spectrograms import MelSpec2Audio, MelSpectrogram


def speechlike(n=16000 * 2, fs=16000, seed=0):
    """Broadband test signal with speech-like slow modulations: a harmonic source plus noise, each octave band with its own 2-8 Hz
    envelope (every 1/3-octave band of the intelligibility model carries modulated energy)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t)
    voiced = sum(np.sin(2 * np.pi * k * np.cumsum(f0) / fs) / k for k in range(1, 40))
    spec = np.fft.rfft(rng.standard_normal(n))
    freqs = np.fft.rfftfreq(n, 1 / fs)
    x = np.zeros(n)
    lo = 100.0
    while lo < fs / 2:
        band = np.where((freqs >= lo) & (freqs < 2 * lo), spec, 0)
        env = 0.55 + 0.45 * np.sin(2 * np.pi * rng.uniform(2, 8) * t + rng.uniform(0, 6.28))
        x += np.fft.irfft(band, n) * env ** 2 * (300.0 / lo) ** 0.5
        lo *= 2
    syll = 0.5 * (1 + np.sin(2 * np.pi * 3.1 * t))
    return ((0.05 * voiced + x) * syll * 0.1).astype(np.float64)


def test_estoi_properties():
    x = speechlike()
    rng = np.random.default_rng(1)
    assert abs(metrics.stoi(x, x, 16000, extended=True) - 1.0) < 1e-6
    assert abs(metrics.stoi(x, x, 16000, extended=False) - 1.0) < 1e-6
    scores = [metrics.stoi(x, x + s * x.std() * rng.standard_normal(len(x)), 16000, extended=True) for s in (0.1, 0.5, 2.0, 8.0)]
    assert all(a > b for a, b in zip(scores, scores[1:])), scores         # monotone in the noise level
    assert scores[0] > 0.9 and scores[-1] < 0.35, scores
    assert abs(metrics.stoi(x, 3.0 * x, 16000, extended=True) - 1.0) < 1e-6       # scale invariant
    with pytest.raises(Exception):
        metrics.stoi(x, x[:-1], 16000)


def test_third_octave_bands_and_silence_removal():
    obm, cf = metrics.thirdoct(10000, 512, 15, 150)
    assert obm.shape == (15, 257) and abs(cf[0] - 150) < 1e-9 and abs(cf[-1] - 150 * 2 ** (14 / 3)) < 1e-6
    assert (obm.sum(axis=0) <= 1).all() and (obm.sum(axis=1) >= 1).all()          # disjoint, non-empty bands
    x = np.concatenate([speechlike(8000), np.zeros(8000), speechlike(8000, seed=2)])
    xs, ys = metrics.remove_silent_frames(x, x, 40, 256, 128)
    assert len(xs) == len(ys) and len(xs) < 0.75 * len(x)


def test_melspec2audio_inverts_the_mel_transform():
    """Vocoding a real signal's log-mel gives a waveform whose log-mel is close to the input (Griffin-Lim recovers a consistent phase;
    InverseMelScale recovers a spectrum with that mel) and which is intelligible against the original by ESTOI."""
    torch.manual_seed(0)
    x = torch.from_numpy(speechlike(16000)).float().unsqueeze(0)
    mel_t, voc = MelSpectrogram(), MelSpec2Audio(max_iters=96)
    mel = mel_t(x)
    y = voc(mel, generator=torch.Generator().manual_seed(0))
    assert y.shape == (1, 256 * (mel.shape[-1] - 1)) and torch.isfinite(y).all()
    mel2 = mel_t(y)
    n = min(mel.shape[-1], mel2.shape[-1])
    loud = mel[..., :n] > mel.max() - 8.0                                           # compare where there is signal
    assert (mel[..., :n] - mel2[..., :n])[loud].abs().mean() < 0.5
    m = min(x.shape[1], y.shape[1])
    assert metrics.stoi(x[0, :m].numpy(), y[0, :m].numpy(), 16000, extended=True) > 0.5


@pytest.mark.gpu
def test_evaluate_net_end_to_end():
    """evaluate.py:22-51 through the boundary: collated batch -> HIP `net(..., tf_ratio=1)[1]` -> vocoder on the device -> ESTOI."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from model.model import get_network
    from lip2speech_amd import callers, synth
    B, T, S = 2, 29, 77
    net = get_network("test").cuda()
    audio = torch.from_numpy(np.stack([speechlike(256 * (S - 1), seed=s) for s in range(B)])).float()
    batch = ((synth.synth_video(B, T, tag="ev"), torch.full((B,), T)), (audio, torch.full((B,), audio.shape[1])),
             (synth.synth_mels(B, S, tag="ev"), torch.full((B,), S), torch.zeros(B, S)), None)

    class Spk:
        def inference(self, a):
            return synth.synth_speaker_embedding(a.shape[0], tag="ev").to(a.device)
    score = callers.evaluate_net(net, [batch], speaker_encoder=Spk(), max_iters=8)
    assert isinstance(score, float) and -1.0 <= score <= 1.0


def test_inverse_mel_device_stopping_rule_and_grouped_calls():
    """`inverse_mel` evaluates torchaudio 0.9's two stopping rules on the device (no host read per iteration) and must end at the iterate
    the host-checked loop ends at; and G loader batches vocoded in one pass (`rows_per_call`) must equal G separate calls."""
    from lip2speech_amd.datasets import MelSpec2Audio
    voc = MelSpec2Audio(max_iters=64)
    g = torch.Generator().manual_seed(3)
    mel = torch.rand(4, 80, 9, generator=g) * 2.0

    def host_loop(m, seed):                      # torchaudio.transforms.InverseMelScale.forward, 0.9.0, with its `.item()` checks
        B, _, L = m.shape
        target = m.transpose(1, 2)
        spec = torch.rand(1, B * L, voc.fb.shape[0], generator=torch.Generator().manual_seed(seed)).reshape(B, L, -1)
        vel, loss, iters = torch.zeros_like(spec), float("inf"), 0
        for _ in range(voc.max_iters):
            diff = target - spec @ voc.fb
            new_loss = float(diff.pow(2).sum(dim=-1).mean())
            vel = 0.9 * vel + (-2.0 / (B * L)) * (diff @ voc.fb.t())
            spec = (spec - 0.1 * vel).clamp_(min=0)
            iters += 1
            if new_loss < 1e-5 or abs(loss - new_loss) < 1e-8:
                break
            loss = new_loss
        return spec.transpose(1, 2), iters
    want, iters = host_loop(mel, 11)
    got = voc.inverse_mel(mel, generator=torch.Generator().manual_seed(11))
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    # two calls of 2 rows in one pass == the two calls on their own (own loss mean, own 1/(B*L), own stop)
    both = torch.cat([mel[:2], mel[2:]], dim=0)
    one_pass = voc.inverse_mel(both, generator=torch.Generator().manual_seed(5), rows_per_call=2)
    gen = torch.Generator().manual_seed(5)
    sep = torch.cat([voc.inverse_mel(mel[:2], generator=gen), voc.inverse_mel(mel[2:], generator=gen)], dim=0)
    assert torch.allclose(one_pass, sep, rtol=1e-5, atol=1e-6)
    assert 1 <= iters <= voc.max_iters
