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


def test_estoi_frame_rule_of_pystoi_033_by_hand():
    """pystoi 0.3.3 frames with range(0, len - framelen, hop) in remove_silent_frames AND stft: the frame ending on the last sample is not
    taken.  Hand-computed at 10 kHz (no resampler), stationary noise (no frame is silent), framelen 256, hop 128:
      len 4096 = 256 + 30 * 128 -> 30 frames kept -> overlap-add (30 - 1) * 128 + 256 = 3968 -> spectra of 29 frames < 30 -> 1e-5
      len 4224 = 256 + 31 * 128 -> 31 kept -> 4096 samples -> 30 frames = exactly one segment -> x against x scores 1
      len 4225 -> ceil(3969 / 128) = 32 kept -> 4224 -> 31 frames -> two segments."""
    assert [metrics.n_frames(n, 256, 128) for n in (256, 257, 384, 385, 4096, 4224, 4225)] == [0, 1, 1, 2, 30, 31, 32]
    x = np.random.default_rng(11).standard_normal(4225)
    assert len(metrics.remove_silent_frames(x[:4096], x[:4096], 40, 256, 128)[0]) == 3968
    assert metrics._stft(np.zeros(3968), 256, 512).shape[0] == 29
    assert metrics.stoi(x[:4096], x[:4096], 10000, extended=True) == 1e-5
    assert abs(metrics.stoi(x[:4224], x[:4224], 10000, extended=True) - 1.0) < 1e-9
    assert abs(metrics.stoi(x, x, 10000, extended=True) - 1.0) < 1e-9


@pytest.mark.gpu
def test_estoi_kernel_frame_rule_by_hand():
    """The same hand-computed cases through `l2s_estoi`: 4096 samples at 10 kHz -> 1e-5 (29 frames), 4224 -> one segment, score 1."""
    x = torch.from_numpy(np.random.default_rng(11).standard_normal((3, 4225)).astype(np.float32)).cuda()
    assert np.allclose(metrics.estoi_device(x[:, :4096].contiguous(), x[:, :4096].contiguous(), 10000).cpu().numpy(), 1e-5)
    assert np.abs(metrics.estoi_device(x[:, :4224].contiguous(), x[:, :4224].contiguous(), 10000).cpu().numpy() - 1.0).max() < 1e-4
    y = x + 0.5 * torch.roll(x, 1, 0)
    want = np.array([metrics.stoi(x[i].cpu().numpy().astype(np.float64), y[i].cpu().numpy().astype(np.float64), 10000, extended=True) for i in range(3)])
    assert np.abs(metrics.estoi_device(x, y.contiguous(), 10000).cpu().numpy() - want).max() < 1e-4


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


def test_resample_plan_and_band_edges_reproduce_scipy_and_thirdoct():
    """The host-side constants handed to `l2s_estoi`: the polyphase FIR / offsets of pystoi's `resample_oct` = scipy.signal.resample_poly
    with pystoi's Octave-compatible window (the kernel computes out[n] = sum_i x[i] h[(n + n_pre_remove) down - i up]) and the one-third
    octave band edges."""
    rng = np.random.default_rng(5)
    # the window itself, by hand for 16 kHz -> 10 kHz (p, q = 5, 8): cut-off 1/16, transition 1/160, 60 dB -> half length ceil(52 / (28.714 / 160)) = 290
    w = metrics.resample_window_oct(10000, 16000)
    assert len(w) == 2 * 290 + 1 and np.argmax(w) == 290 and abs(w[290] - 2 * 5 / 16) < 1e-12 and np.allclose(w, w[::-1])
    for n_in, fs in ((19456, 16000), (4001, 16000), (3000, 22050), (5000, 8000)):
        x = rng.standard_normal(n_in)
        h, up, down, n_pre, n_out = metrics.resample_poly_plan(n_in, metrics.FS, fs)
        want = metrics.resample_oct(x, metrics.FS, fs)
        assert len(want) == n_out
        got = np.zeros(n_out)
        for n in range(0, n_out, 7):                     # every 7th output, by the kernel's formula
            c = (n + n_pre) * down
            i = np.arange(max(0, -(-(c - (len(h) - 1)) // up)), min(n_in - 1, c // up) + 1)
            got[n] = np.dot(x[i], h[c - i * up])
        assert np.abs(got[::7] - want[::7]).max() < 1e-12
    assert metrics.resample_poly_plan(1000, 10000, 10000)[0] is None
    be = metrics.band_edges()
    obm, _ = metrics.thirdoct(metrics.FS, metrics.NFFT, metrics.NUMBAND, metrics.MINFREQ)
    for b in range(15):
        row = np.zeros(257)
        row[be[b]:be[15 + b]] = 1
        assert np.array_equal(row, obm[b])
    assert all(be[b] <= be[b + 1] for b in range(14))


@pytest.mark.gpu
@pytest.mark.parametrize("N,L,R,iters", [(4, 77, 2, 256), (3, 20, 3, 40), (2, 77, 1, 0)])
def test_inverse_mel_kernel_matches_the_torch_restatement(N, L, R, iters):
    """`l2s_inverse_mel` (one wave per mel frame, banded filterbank, stopping rules on the device) against `MelSpec2Audio.inverse_mel` on
    torch ops, same start iterate: every spectrum value, the per-iteration losses, and the iteration count the stopping rules leave."""
    from lip2speech_amd import native
    voc_t = MelSpec2Audio(max_iters=iters, backend="torch").cuda()
    voc_h = MelSpec2Audio(max_iters=iters, backend="hip").cuda()
    g = torch.Generator(device="cuda")
    mel = torch.exp(torch.randn(N, 80, L, device="cuda", generator=g.manual_seed(1)) * 2.0 - 5.0)
    want = voc_t.inverse_mel(mel, g.manual_seed(7), rows_per_call=R)
    got = voc_h.inverse_mel(mel, g.manual_seed(7), rows_per_call=R)
    assert got.shape == want.shape == (N, 513, L)
    assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    if iters:
        init = torch.rand(N // R, R * L, 513, device="cuda", generator=g.manual_seed(7)).reshape(N * L, 513)
        spec, loss, ran = native.inverse_mel(mel, voc_h.fb, voc_h.fb_nnz, init, iters, rows_per_call=R, want_loss=True)
        assert torch.equal(spec, got)
        assert ran.dtype == torch.int32 and int(ran.min()) >= 1 and int(ran.max()) <= iters
        # the loss of iteration 0 is the start iterate's: mean over the call's rows of |mel - init @ fb|^2
        l0 = (mel.transpose(1, 2).reshape(N // R, R * L, 80) - init.reshape(N // R, R * L, 513) @ voc_h.fb).pow(2).sum(-1).mean(-1)
        assert torch.allclose(loss[:, 0], l0, rtol=1e-4)
    # log-mel input: exp applied on the fly
    got_log = native.inverse_mel(torch.log(mel), voc_h.fb, voc_h.fb_nnz,
                                 torch.rand(N // R, R * L, 513, device="cuda", generator=g.manual_seed(7)).reshape(N * L, 513), iters, rows_per_call=R, log_input=True)
    assert float((got_log - want).abs().max()) < 2e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("N,L,iters,tol", [(3, 77, 0, 2e-5), (3, 77, 1, 5e-5), (2, 77, 4, 1e-4), (2, 40, 8, 2e-4), (1, 100, 2, 1e-4)])
def test_griffin_lim_kernel_matches_the_torch_restatement(N, L, iters, tol):
    """`l2s_griffin_lim` (one block per clip, waveform in LDS, radix-8 wave FFTs, deterministic four-phase overlap-add) against
    `MelSpec2Audio.griffin_lim` on torch.stft / istft, same start angles (a C2R transform ignores the imaginary parts of the DC and
    Nyquist bins; the start angles get real ones so that the comparison does not hinge on that convention)."""
    from lip2speech_amd import native
    g = torch.Generator(device="cuda")
    power = torch.rand(N, 513, L, device="cuda", generator=g.manual_seed(2)) ** 4 * 3.0
    ang = torch.rand(N, 513, L, 2, device="cuda", generator=g.manual_seed(3))
    ang[:, 0, :, 1] = 0
    ang[:, 512, :, 1] = 0
    voc = MelSpec2Audio(max_iters=iters, backend="torch").cuda()
    # the restatement with explicit angles
    mag = power.sqrt()
    a = torch.view_as_complex(ang.clone())
    prev = torch.zeros_like(a)
    length = 256 * (L - 1)
    stft = lambda x: torch.stft(x, 1024, 256, 1024, voc.window, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)   # noqa: E731
    istft = lambda z: torch.istft(z, 1024, 256, 1024, voc.window, length=length)          # noqa: E731
    for _ in range(iters):
        rebuilt = stft(istft(mag * a))
        a = rebuilt - prev * (0.99 / 1.99)
        a = a / (a.abs() + 1e-16)
        prev = rebuilt
    want = istft(mag * a)
    got = native.griffin_lim(power, ang, iters)
    assert got.shape == want.shape == (N, length)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < tol * max(1.0, scale), (float((got - want).abs().max()), scale)


@pytest.mark.gpu
def test_griffin_lim_kernel_full_length_run_converges_like_the_restatement():
    """256 iterations (evaluate.py's setting): phase retrieval amplifies rounding differences between two FFT implementations, so the
    two waveforms are compared through what Griffin-Lim optimises - the spectral inconsistency |  |stft(y)| - mag | / |mag| - and through
    their magnitude spectrograms."""
    from lip2speech_amd import native
    g = torch.Generator(device="cuda")
    x = torch.from_numpy(np.stack([speechlike(256 * 76, seed=s) for s in range(2)])).float().cuda()
    win = torch.hann_window(1024, periodic=True, device="cuda")
    stft = lambda v: torch.stft(v, 1024, 256, 1024, win, center=True, pad_mode="reflect", return_complex=True)       # noqa: E731
    power = stft(x).abs() ** 2
    ang = torch.rand(2, 513, 77, 2, device="cuda", generator=g.manual_seed(3))
    voc = MelSpec2Audio(max_iters=256, backend="torch").cuda()
    got = native.griffin_lim(power, ang, 256)
    torch.manual_seed(0)
    want = voc.griffin_lim(power, g.manual_seed(3))
    mag = power.sqrt()
    inc = lambda y: float(((stft(y).abs() - mag).norm() / mag.norm()))      # noqa: E731
    assert torch.isfinite(got).all()
    assert inc(got) < 1.15 * inc(want) + 1e-3, (inc(got), inc(want))
    assert inc(got) < 0.35


@pytest.mark.gpu
def test_estoi_kernel_matches_the_numpy_restatement_on_sample_lrw():
    """`l2s_estoi` per clip against `metrics.stoi(., ., 16000, extended=True)` (fp64 numpy) on the ten SAMPLE_LRW clips' audio: degraded
    copies at four noise levels, a vocoded copy, and the clean signal itself (-> 1)."""
    import os
    sample = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_lrw")
    clean = np.stack([np.load(os.path.join(sample, f"ABOUT_{i:05d}.npz"))["data"] for i in range(1, 11)]).astype(np.float32)
    assert clean.shape == (10, 19456)
    rng = np.random.default_rng(0)
    for level in (0.0, 0.1, 0.5, 2.0, 8.0):
        pred = (clean + level * clean.std(axis=1, keepdims=True) * rng.standard_normal(clean.shape)).astype(np.float32)
        want = np.array([metrics.stoi(clean[i], pred[i], 16000, extended=True) for i in range(10)])
        got = metrics.estoi_device(torch.from_numpy(clean).cuda(), torch.from_numpy(pred).cuda(), 16000).cpu().numpy()
        assert np.abs(got - want).max() < 1e-4, (level, got, want)
    # a clip too short for one segment: pystoi's 1e-5
    short = torch.from_numpy(clean[:, :4000]).cuda()
    assert np.allclose(metrics.estoi_device(short, short, 16000).cpu().numpy(), 1e-5)
    assert metrics.stoi(clean[0, :4000], clean[0, :4000], 16000, extended=True) == 1e-5
    # already at 10 kHz: no resampler
    x10 = torch.from_numpy(clean[:, :12160]).cuda()
    y10 = x10 + 0.3 * x10.std() * torch.randn(x10.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    want = np.array([metrics.stoi(x10[i].cpu().numpy(), y10[i].cpu().numpy(), 10000, extended=True) for i in range(10)])
    assert np.abs(metrics.estoi_device(x10, y10, 10000).cpu().numpy() - want).max() < 1e-4


@pytest.mark.gpu
def test_evaluate_net_device_tail_matches_host_tail():
    """evaluate.py:22-51 with the vocoder and the metric on the device against the same run scored on the host (numpy ESTOI of the same
    device-vocoded waveforms), and the hip vocoder against the torch one through ESTOI (both start from the same random iterates)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from model.model import get_network
    from lip2speech_amd import callers, synth
    B, T, S = 4, 29, 77
    net = get_network("test").cuda()
    audio = torch.from_numpy(np.stack([speechlike(256 * (S - 1), seed=s) for s in range(B)])).float()
    mel_t = MelSpectrogram()
    batches = [((synth.synth_video(B, T, tag=f"ev{i}"), torch.full((B,), T)), (audio, torch.full((B,), audio.shape[1])),
                (mel_t(audio)[:, :, :S], torch.full((B,), S), torch.zeros(B, S)), None) for i in range(3)]

    class Spk:
        def inference(self, a):
            return synth.synth_speaker_embedding(a.shape[0], tag="ev").to(a.device)
    scores = {}
    for vb, me in (("hip", "hip"), ("hip", "host"), ("torch", "host")):
        torch.manual_seed(0)
        scores[vb, me] = callers.evaluate_net(net, batches, speaker_encoder=Spk(), max_iters=32, vocoder_backend=vb, metric=me)
    assert abs(scores["hip", "hip"] - scores["hip", "host"]) < 1e-4, scores
    assert abs(scores["hip", "host"] - scores["torch", "host"]) < 5e-3, scores
