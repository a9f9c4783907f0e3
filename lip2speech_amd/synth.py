"""Deterministic synthetic weights and inputs for the hot path.

There is no trained checkpoint and no network, so every test, the golden
generator and ``bench.py`` regenerate the same weights and inputs from integers
only: a splitmix64 counter hash keyed by the tensor's checkpoint key, top 24 bits
scaled by 2**-24 (exact in fp32).  No libm call is involved, so the values are
bit-identical in the build container and on the GPU box.

Scales follow the reference's initialisers so the synthetic network sits in the
same numeric regime:
  * encoder convs  N(0, sqrt(2/(k*Cout)))   reference/model/modules/video.py:27-43
    -> variance-matched uniform
  * LinearNorm/ConvNorm  xavier-uniform with gain   reference/model/modules/decoder.py:78-80,99-100
  * plain nn.Linear/nn.Conv1d  U(+-1/sqrt(fan_in)); nn.LSTM  U(+-1/sqrt(hidden))
BatchNorm running statistics, PReLU slopes and PSine ``w`` are randomised on
purpose: their defaults (0/1/0.25/1) would hide BN-folding and activation bugs.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

from . import statespec

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int, seed: int = 1234) -> np.ndarray:
    """n values in [0,1) on the 2**-24 grid, fp32-exact; stream identified by (key, seed)."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _MASK
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def sym_uniform(key: str, shape, bound: float, seed: int = 1234) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(key, n, seed)
    return ((u - np.float32(0.5)) * np.float32(2.0 * bound)).reshape(shape)


def pseudo_normal(key: str, shape, sigma: float = 1.0, seed: int = 1234) -> np.ndarray:
    """Variance-sigma**2 'normal': Irwin-Hall sum of four uniforms (adds/mults only, no libm)."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float32)
    for j in range(4):
        acc += uniform01(f"{key}#{j}", n, seed)
    out = (acc - np.float32(2.0)) * np.float32(sigma * math.sqrt(3.0))
    return out.reshape(shape)


_GAIN = {"linear": 1.0, "sigmoid": 1.0, "tanh": 5.0 / 3.0}


def positional_table(n_position: int = statespec.MAX_STEPS, d_hid: int = statespec.D_MODEL) -> np.ndarray:
    """decoder.positional_encodings.pos_table - float64 sin/cos then cast
    (reference/model/modules/decoder.py:19-40).  The committed goldens pin the exact bits."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    angle = pos / np.power(10000.0, 2 * (j // 2) / d_hid)[None, :]
    table = angle.copy()
    table[:, 0::2] = np.sin(angle[:, 0::2])
    table[:, 1::2] = np.cos(angle[:, 1::2])
    return table.astype(np.float32)[None]


def _draw(key: str, shape: Tuple[int, ...], kind: str, seed: int) -> torch.Tensor:
    if kind == "bn_nbt":
        return torch.zeros((), dtype=torch.int64)
    if kind == "pos_table":
        return torch.from_numpy(positional_table(shape[1], shape[2]))
    if kind == "conv_enc":
        n = int(np.prod(shape[2:])) * shape[0]
        sigma = math.sqrt(2.0 / n)
        arr = sym_uniform(key, shape, sigma * math.sqrt(3.0), seed)
    elif kind == "bn_w":
        arr = 0.5 + uniform01(key, shape[0], seed)
    elif kind == "bn_b":
        arr = sym_uniform(key, shape, 0.2, seed)
    elif kind == "bn_rm":
        arr = sym_uniform(key, shape, 0.2, seed)
    elif kind == "bn_rv":
        arr = 0.5 + uniform01(key, shape[0], seed)
    elif kind == "prelu":
        arr = 0.1 + 0.3 * uniform01(key, shape[0], seed)
    elif kind == "psine":
        arr = 0.5 + uniform01(key, shape[0], seed)
    elif kind.startswith("xavier:"):
        gain = _GAIN[kind.split(":")[1]]
        recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * recept, shape[0] * recept
        arr = sym_uniform(key, shape, gain * math.sqrt(6.0 / (fan_in + fan_out)), seed)
    elif kind == "default":
        recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        arr = sym_uniform(key, shape, 1.0 / math.sqrt(shape[1] * recept), seed)
    elif kind.startswith("bias:"):
        arr = sym_uniform(key, shape, 1.0 / math.sqrt(int(kind.split(":")[1])), seed)
    elif kind.startswith("lstm:"):
        arr = sym_uniform(key, shape, 1.0 / math.sqrt(int(kind.split(":")[1])), seed)
    elif kind == "bos":
        arr = pseudo_normal(key, shape, 1.0, seed)
    elif kind.startswith("temp:"):
        # learned scalar, init sqrt(d) (decoder.py:236,304); perturbed so it is not a round number
        d = int(kind.split(":")[1])
        arr = np.asarray([math.sqrt(d)], dtype=np.float32) * (0.9 + 0.2 * uniform01(key, 1, seed))
    elif kind == "emb":
        arr = uniform01(key, int(np.prod(shape)), seed).reshape(shape)
    else:
        raise ValueError(f"unknown kind {kind!r} for {key}")
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))


def synth_state_dict(spec: Iterable = None, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint for the measured path (encoder.* + decoder.* keys)."""
    spec = statespec.model_spec() if spec is None else spec
    return {key: _draw(key, tuple(shape), kind, seed) for key, shape, kind in spec}


def synth_video(B: int, T: int, H: int = 96, W: int = 96, seed: int = 1234, tag: str = "video") -> torch.Tensor:
    """(B,3,T,H,W) fp32, unit variance - real frames are ImageNet-normalised
    (reference/datasets/lrw/dataset.py:83-86), so unit-normal-like is representative."""
    return torch.from_numpy(pseudo_normal(f"{tag}:{B}x{T}x{H}x{W}", (B, 3, T, H, W), 1.0, seed))


def synth_speaker_embedding(B: int, seed: int = 1234, tag: str = "spk") -> torch.Tensor:
    """normalize(relu(N(0,1))) like SpeakerEncoder.inference (reference/model/modules/audio.py:144-150);
    the L2 norm is taken in float64 and rounded once."""
    raw = np.maximum(pseudo_normal(f"{tag}:{B}", (B, statespec.D_EMB), 1.0, seed), 0.0).astype(np.float64)
    raw /= np.maximum(np.sqrt((raw * raw).sum(axis=1, keepdims=True)), 1e-12)
    return torch.from_numpy(raw.astype(np.float32))


def synth_gumbel(rows: int, seed: int = 1234, tag: str = "gumbel") -> torch.Tensor:
    """Gumbel(0,1) noise -log(-log(u)) computed in float64 then rounded to fp32.  Transcendental,
    so fixtures that must be bit-stable across hosts are committed under tests/golden/ instead
    of being regenerated; bench.py only needs representative noise."""
    u = uniform01(f"{tag}:{rows}", rows * statespec.VOCAB, seed).astype(np.float64)
    u = np.clip(u, 2.0 ** -24, 1.0 - 2.0 ** -24)
    g = -np.log(-np.log(u))
    return torch.from_numpy(g.astype(np.float32).reshape(rows, statespec.VOCAB))


def synth_mels(B: int, S: int, seed: int = 1234, tag: str = "mel") -> torch.Tensor:
    """Target mels ~ N(-5,2) clipped at ln(1e-5) (the collate pad value,
    reference/datasets/__init__.py:17)."""
    m = pseudo_normal(f"{tag}:{B}x{S}", (B, statespec.N_MELS, S), 2.0, seed) - np.float32(5.0)
    return torch.from_numpy(np.maximum(m, np.float32(-11.5129)).astype(np.float32))


def synth_clip_lengths(B: int, lo: int, hi: int, tag: str) -> np.ndarray:
    """Deterministic per-clip frame counts in [lo, hi] (variable-length corpora: GRID up to 75 frames, AVSpeech 25-50); the clip with the
    largest draw is forced to `hi` so the padded batch has T = hi."""
    u = uniform01(f"lens:{tag}", B)
    t = (lo + np.floor(u.astype(np.float64) * (hi - lo + 1))).astype(np.int64).clip(lo, hi)
    t[int(np.argmax(u))] = hi
    return t


def synth_padded_video(B: int, lens, tag: str) -> torch.Tensor:
    """What the collate hands the model for clips of different lengths: every clip zero-padded to the batch maximum
    (reference/datasets/__init__.py:17,29-31).  The model ignores lengths, so the padding is part of the semantics."""
    T = int(np.max(lens))
    video = synth_video(B, T, tag=tag)
    for b, t in enumerate(lens):
        video[b, :, int(t):] = 0
    return video


def synth_audio(B: int, n_samples: int, tag: str, sigma: float = 0.1) -> torch.Tensor:
    """(B, n_samples) noise-like 16 kHz audio in about +-0.35 (the range of the LRW sample clips)."""
    return torch.from_numpy(pseudo_normal(f"audio:{tag}", (B, n_samples), sigma))
