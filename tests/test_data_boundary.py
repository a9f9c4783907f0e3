"""Data boundary (collates, mel transform, LRW sample loader) - CPU."""
import os

import numpy as np
import torch

from lip2speech_amd.datasets import LRW, MelSpectrogram, test_collate_fn_pad, train_collate_fn_pad
from lip2speech_amd.datasets.lrw import load_frames, normalise_mouth

SAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_lrw")


def _item(T, N, M, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(T, 3, 96, 96, generator=g), torch.randn(1, N, generator=g), torch.randn(80, M, generator=g), torch.randn(2, 3, 160, 160, generator=g))


def test_collate_layouts_and_padding():
    batch = [_item(25, 16000, 63, 0), _item(29, 19456, 77, 1)]
    (video, vlen), (audio, alen), (mels, mlen, gate), faces = train_collate_fn_pad(batch)
    assert video.shape == (2, 3, 29, 96, 96) and vlen.tolist() == [25, 29]
    assert torch.equal(video[0, :, 25:], torch.zeros(3, 4, 96, 96))                 # zero frames after the clip end
    assert torch.equal(video[0, :, :25], batch[0][0].permute(1, 0, 2, 3))
    assert audio.shape == (2, 19456) and alen.tolist() == [16000, 19456] and audio[0, 16000:].abs().sum() == 0
    assert mels.shape == (2, 80, 77) and torch.all(mels[0, :, 63:] == -11.5129)
    assert gate[0].tolist() == [0.0] * 62 + [1.0] * 15 and gate[1].tolist() == [0.0] * 76 + [1.0]
    assert faces.shape == (2, 2, 3, 160, 160)
    out = test_collate_fn_pad([b + (("f", "a"),) for b in batch])
    assert len(out) == 5 and out[4] == (("f", "a"), ("f", "a"))


def test_mel_transform_against_direct_dft():
    """80-mel / n_fft 1024 / hop 256 log-mel: check torch.stft route against an explicit DFT of one frame."""
    mel = MelSpectrogram()
    t = torch.arange(19456) / 16000.0
    wav = (0.3 * torch.sin(2 * np.pi * 440 * t) + 0.1 * torch.sin(2 * np.pi * 3000 * t)).unsqueeze(0)
    out = mel(wav)
    assert out.shape == (1, 80, 19456 // 256 + 1) and out.min() >= np.log(1e-5) - 1e-6
    # frame 10 by hand: reflect-padded signal, hann window, |DFT|^2, triangular HTK filters
    padded = torch.nn.functional.pad(wav.unsqueeze(0), (512, 512), mode="reflect")[0, 0]
    frame = padded[10 * 256: 10 * 256 + 1024].double() * torch.hann_window(1024, periodic=True).double()
    k = torch.arange(513).double().unsqueeze(1) * torch.arange(1024).double().unsqueeze(0) * (2 * np.pi / 1024)
    power = (torch.cos(k) @ frame) ** 2 + (torch.sin(k) @ frame) ** 2
    want = torch.log(torch.clamp(power.float() @ mel.fb, min=1e-5))
    assert (out[0, :, 10] - want).abs().max() < 1e-3
    assert out[0, :, 10].argmax() == want.argmax()
    # speaker-encoder front-end shape: 40 mels, n_fft 400, hop 160, no log (audio.py:121)
    m40 = MelSpectrogram(n_fft=400, hop_length=160, win_length=400, n_mels=40, f_min=0.0, f_max=8000.0, log=False)
    assert m40(wav).shape == (1, 40, 19456 // 160 + 1)


def test_lrw_sample_clips_load():
    frames = load_frames(os.path.join(SAMPLE, "ABOUT_00001_mouth.npz"))
    assert frames.shape == (29, 96, 96, 3) and frames.dtype == np.uint8
    x = normalise_mouth(frames)
    assert x.shape == (29, 3, 96, 96) and abs(float(x.mean())) < 3
    audio = np.load(os.path.join(SAMPLE, "ABOUT_00001.npz"))["data"]
    assert audio.shape == (19456,) and audio.dtype == np.float32
    assert MelSpectrogram()(torch.from_numpy(audio[None])).shape == (1, 80, 77)     # LRW: S = 77


def test_lrw_dataset_index_rebuild(tmp_path):
    # lay the two fixture clips out like the reference's tree; the CSV index is absent there too
    import shutil
    d = tmp_path / "LRW_Faces" / "ABOUT" / "test"
    a = tmp_path / "lipread_audio" / "ABOUT" / "test"
    d.mkdir(parents=True); a.mkdir(parents=True)
    for i in (1, 2):
        shutil.copy(os.path.join(SAMPLE, f"ABOUT_0000{i}_mouth.npz"), d / f"ABOUT_0000{i}_mouth.npz")
        shutil.copy(os.path.join(SAMPLE, f"ABOUT_0000{i}.npz"), a / f"ABOUT_0000{i}.npz")
    ds = LRW(str(tmp_path), mode="test")
    assert len(ds) == 2
    (video, vlen), (audio, alen), (mels, mlen, gate), faces = train_collate_fn_pad([ds[0], ds[1]])
    assert video.shape == (2, 3, 29, 96, 96) and mels.shape == (2, 80, 77) and vlen.tolist() == [29, 29]


def test_device_collate_packs_raw_clips(tmp_path):
    """`LRW(raw_frames=True)` + `device_collate_fn_pad`: the uint8 clips travel packed (a quarter of the bytes of the fp32 batch); audio,
    mels, gate and lengths are exactly what `train_collate_fn_pad` builds.  (The kernel that finishes the job is GPU-tested.)"""
    import shutil
    from lip2speech_amd.datasets import PackedFrames, device_collate_fn_pad
    d = tmp_path / "LRW_Faces" / "ABOUT" / "test"
    a = tmp_path / "lipread_audio" / "ABOUT" / "test"
    d.mkdir(parents=True); a.mkdir(parents=True)
    for i in (1, 2):
        shutil.copy(os.path.join(SAMPLE, f"ABOUT_0000{i}_mouth.npz"), d / f"ABOUT_0000{i}_mouth.npz")
        shutil.copy(os.path.join(SAMPLE, f"ABOUT_0000{i}.npz"), a / f"ABOUT_0000{i}.npz")
    raw, ref = LRW(str(tmp_path), mode="test", raw_frames=True), LRW(str(tmp_path), mode="test")
    assert raw[0][0].dtype == torch.uint8 and raw[0][0].shape == (29, 96, 96, 3)
    (packed, vlen), (audio, alen), (mels, mlen, gate), faces = device_collate_fn_pad([raw[0], raw[1]])
    (video, vlen2), (audio2, alen2), (mels2, mlen2, gate2), faces2 = train_collate_fn_pad([ref[0], ref[1]])
    assert isinstance(packed, PackedFrames) and packed.frames == [29, 29] and packed.offsets == [0, 29 * 96 * 96 * 3]
    assert packed.data.numel() == video.numel() and packed.data.element_size() * 4 == video.element_size()      # 1 byte per value instead of 4
    assert torch.equal(vlen, vlen2) and torch.equal(audio, audio2) and torch.equal(mels, mels2) and torch.equal(gate, gate2) and torch.equal(mlen, mlen2)
    # the packed bytes are the decoded frames, clip after clip
    assert torch.equal(packed.data[:29 * 96 * 96 * 3].view(29, 96, 96, 3), raw[0][0])
    # ragged clips: offsets stay 4-byte aligned
    p2 = PackedFrames([torch.zeros(3, 6, 6, 3, dtype=torch.uint8), torch.ones(5, 6, 6, 3, dtype=torch.uint8)], pin=False)
    assert p2.frames == [3, 5] and p2.offsets == [0, 324] and p2.lengths.tolist() == [3, 5]


def test_lrw_face_crops_and_rng_consumption(tmp_path):
    """datasets/lrw/dataset.py:139-141: every item draws `torch.rand(2)` face-frame indices and resizes those frames to 160x160 for the
    (third-party) face tower.  An epoch's RNG consumption must equal the reference's whether or not the face file is there."""
    import bz2
    import io
    import pickle
    import shutil
    from PIL import Image
    d = tmp_path / "LRW_Faces" / "ABOUT" / "test"
    a = tmp_path / "lipread_audio" / "ABOUT" / "test"
    d.mkdir(parents=True); a.mkdir(parents=True)
    for i in (1, 2):
        shutil.copy(os.path.join(SAMPLE, f"ABOUT_0000{i}_mouth.npz"), d / f"ABOUT_0000{i}_mouth.npz")
        shutil.copy(os.path.join(SAMPLE, f"ABOUT_0000{i}.npz"), a / f"ABOUT_0000{i}.npz")
    # a face file in the reference's format (bz2 pickle of JPEG byte arrays, variable size) for clip 1 only: frame f is a flat grey level 8*f
    blobs = []
    for f in range(29):
        buf = io.BytesIO()
        Image.fromarray(np.full((146, 120, 3), 8 * f, dtype=np.uint8)).save(buf, format="JPEG", quality=95)
        blobs.append(np.frombuffer(buf.getvalue(), dtype=np.uint8).reshape(-1, 1))
    with bz2.BZ2File(str(d / "ABOUT_00001_face.npz"), "w") as fh:
        pickle.dump(blobs, fh)
    ds = LRW(str(tmp_path), (96, 96), "test", False, 1, None, "stray", extra="ignored")   # stray positional / keyword arguments do not break construction
    torch.manual_seed(5)
    want = (torch.rand(2) * 29).int().tolist()
    after_two = torch.rand(1)
    torch.manual_seed(5)
    after_four = (torch.rand(2), torch.rand(2), torch.rand(1))[2]
    torch.manual_seed(5)
    with_face = ds[0][3]
    assert torch.equal(torch.rand(1), after_two)                    # exactly two draws, like the reference
    assert with_face.shape == (2, 3, 160, 160)
    for k, f in enumerate(want):                                    # the drawn frames, resized, then (x - 127.5) / 128
        assert abs(float(with_face[k].mean()) - (8 * f - 127.5) / 128.0) < 2 / 128.0
    torch.manual_seed(5)
    ds[0]; no_face = ds[1][3]                                       # clip 2 has no face file: zeros, but the draw is still consumed
    assert torch.equal(no_face, torch.zeros(2, 3, 160, 160)) and torch.equal(torch.rand(1), after_four)


def test_per_corpus_collate_pads_mels_with_zeros():
    """datasets.{grid,avspeech,wild}.av_speech_collate_fn_pad (datasets/grid/dataset.py:28-68): zero-padded mel targets, unlike the top-level
    collate's ln(1e-5); av_speech_collate_fn_trim trims to the batch minimum (datasets/avspeech/dataset.py:30-50)."""
    from datasets.avspeech import av_speech_collate_fn_pad, av_speech_collate_fn_trim
    from datasets.grid import av_speech_collate_fn_pad as grid_pad
    assert grid_pad is av_speech_collate_fn_pad
    batch = [_item(25, 16000, 63, 0), _item(29, 19456, 77, 1)]
    (video, vlen), (audio, alen), (mels, mlen, gate), faces = av_speech_collate_fn_pad(batch)
    ref = train_collate_fn_pad(batch)
    assert torch.equal(video, ref[0][0]) and torch.equal(audio, ref[1][0]) and torch.equal(gate, ref[2][2])
    assert torch.all(mels[0, :, 63:] == 0) and torch.equal(mels[:, :, :63], ref[2][0][:, :, :63])
    (frames, flen), (speech, slen), faces = av_speech_collate_fn_trim(batch)
    assert frames.shape == (2, 25, 3, 96, 96) and flen == [25, 25] and speech.shape == (2, 1, 16000) and slen == [16000, 16000]
