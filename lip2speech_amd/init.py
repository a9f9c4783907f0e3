"""Initial values of a freshly constructed model, family by family as the reference initialises them, drawn from torch's global RNG
(so `torch.manual_seed(hparams.seed)` / train.py:7 `torch.manual_seed(1)` decide them, as in the reference):

  conv_enc   encoder convs: N(0, sqrt(2 / (prod(kernel_size) * out_channels)))        video.py:27-43 (_initialize_weights_randomly)
  bn_*       BatchNorm weight 1, bias 0, running_mean 0, running_var 1, counter 0    video.py:44-46 + nn.BatchNorm defaults
  prelu      0.25 (nn.PReLU default)                                                  video.py:66
  psine      1 (PSine(dims, w=1))                                                     decoder.py:43-49
  xavier:g   LinearNorm / ConvNorm: xavier_uniform_(gain = calculate_gain(g))         decoder.py:78-80, 99-100
  default    nn.Linear / nn.Conv1d weight: kaiming_uniform_(a = sqrt(5)) = U(+-1/sqrt(fan_in));  bias:<fan_in> the matching bias
  lstm:<h>   nn.LSTM: U(+-1/sqrt(hidden))
  bos        randn (decoder.py:289);  temp:<d>  sqrt(d) (decoder.py:237,302);  emb  rand (decoder.py:206)
  pos_table  the sinusoid table (decoder.py:19-40)

`synth.py` keeps its own deliberately randomised families (BatchNorm statistics, PReLU / PSine weights away from their defaults) for tests,
goldens and bench.py, which load `synth.synth_state_dict()` explicitly.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

_GAIN = {"linear": "linear", "sigmoid": "sigmoid", "tanh": "tanh"}


def reference_init(shape: Tuple[int, ...], kind: str) -> torch.Tensor:
    shape = tuple(shape)
    if kind == "bn_nbt":
        return torch.zeros((), dtype=torch.int64)
    if kind == "pos_table":
        from .synth import positional_table
        return torch.from_numpy(positional_table(shape[1], shape[2]))
    if kind in ("bn_w", "bn_rv", "psine"):
        return torch.ones(shape)
    if kind in ("bn_b", "bn_rm"):
        return torch.zeros(shape)
    if kind == "prelu":
        return torch.full(shape, 0.25)
    if kind == "conv_enc":
        n = math.prod(shape[2:]) * shape[0]
        return torch.empty(shape).normal_(0.0, math.sqrt(2.0 / n))
    if kind.startswith("xavier:"):
        w = torch.empty(shape)
        torch.nn.init.xavier_uniform_(w, gain=torch.nn.init.calculate_gain(_GAIN[kind.split(":")[1]]))
        return w
    if kind == "default":
        fan_in = shape[1] * (math.prod(shape[2:]) if len(shape) > 2 else 1)
        b = 1.0 / math.sqrt(fan_in)
        return torch.empty(shape).uniform_(-b, b)
    if kind.startswith(("bias:", "lstm:")):
        b = 1.0 / math.sqrt(int(kind.split(":")[1]))
        return torch.empty(shape).uniform_(-b, b)
    if kind == "bos":
        return torch.randn(shape)
    if kind.startswith("temp:"):
        return torch.ones(shape) * (int(kind.split(":")[1]) ** 0.5)
    if kind == "emb":
        return torch.rand(shape)
    raise ValueError(f"unknown kind {kind!r}")
