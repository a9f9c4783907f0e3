"""CPU ORACLE for the Lip2Speech hot path.  TEST INFRASTRUCTURE - NOT THE PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file.  The shipped path (``lip2speech_amd``) never does: it runs the
HIP kernels behind the C-ABI in ``include/l2s.h`` and raises if that library is
missing.

What it is: a plain fp32 (or fp64, for noise-floor studies) restatement, in
explicit tensor algebra on the CPU, of the eval-mode arithmetic of the reference's

  * ``VideoExtractor.forward``        /root/reference/model/modules/video.py:76-87
  * ``InvertedResidual`` / shuffle    /root/reference/model/modules/shufflenetv2.py:26-104
  * ``Lip2Speech.inference`` glue     /root/reference/model/model.py:43-59
  * ``Decoder.inference``             /root/reference/model/modules/decoder.py:382-444
  * ``Decoder.forward`` (eval, tf=1)  /root/reference/model/modules/decoder.py:320-379
  * ``Postnet`` / ``MultiHopConv`` / ``Content`` / ``PSine``   decoder.py:43-271

It takes a checkpoint-style ``dict`` of tensors (the reference's own key names) so
the very same weights drive the reference modules, this oracle and the HIP path.

Pinning: the reference ships no tests or golden vectors.  This
oracle is pinned against the reference itself, imported in the build container by
``tests/golden/make_goldens.py``; the resulting vectors are committed under
``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks the oracle against
them on every run (no /root/reference needed).  Un-importable pieces
(FaceRecognizer, torchaudio mel front-end, vocoder/ESTOI) are "parity unpinned"
and are not restated here.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5


# ----------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------
class batch_statistics:
    """Context manager: BatchNorm layers behave as in nn.Module.train() (the reference's nn.BatchNorm{1,2,3}d at video.py:70,
    shufflenetv2.py:46-69, decoder.py:118-140,166-226 under train.py:150 `net.train()`): normalise with this batch's mean and biased
    variance; the running statistics a real module would hold afterwards (momentum 0.1, unbiased variance) are recorded in
    ``updates[prefix] = (running_mean, running_var)`` instead of being written into ``sd``."""
    active = None

    def __init__(self, momentum: float = 0.1):
        self.momentum, self.updates = momentum, {}

    def __enter__(self):
        batch_statistics.active = self
        return self

    def __exit__(self, *a):
        batch_statistics.active = None


def batchnorm_eval(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """BatchNorm{1,2,3}d over channel dim 1: (x-mu)/sqrt(var+eps)*gamma+beta with the running statistics (eval), or with the batch's
    own statistics inside a ``batch_statistics()`` block."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    ctx = batch_statistics.active
    if ctx is not None:
        dims = [d for d in range(x.dim()) if d != 1]
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=dims)
        var_b = x.var(dim=dims, unbiased=False)
        m = ctx.momentum
        with torch.no_grad():
            ctx.updates[prefix] = ((1 - m) * sd[prefix + ".running_mean"] + m * mean.detach(),
                                   (1 - m) * sd[prefix + ".running_var"] + m * var_b.detach() * n / (n - 1))
        g = sd[prefix + ".weight"].view(shape)
        b = sd[prefix + ".bias"].view(shape)
        return (x - mean.view(shape)) / torch.sqrt(var_b.view(shape) + BN_EPS) * g + b
    mu = sd[prefix + ".running_mean"].view(shape)
    var = sd[prefix + ".running_var"].view(shape)
    g = sd[prefix + ".weight"].view(shape)
    b = sd[prefix + ".bias"].view(shape)
    return (x - mu) / torch.sqrt(var + BN_EPS) * g + b


def psine_channels_first(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """PSine on a (B,C,T) map: sin(x)*w[c]   (decoder.py:43-70; the permute there only moves C last)."""
    return torch.sin(x) * w.view(1, -1, 1)


def psine_last(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """PSine on (...,C): sin(x)*w."""
    return torch.sin(x) * w


def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def linear(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """PyTorch gate order i,f,g,o (Appendix A of SURVEY.md)."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def channel_shuffle2(x: torch.Tensor) -> torch.Tensor:
    """new[2k+g] = old[g*C/2+k]  (shufflenetv2.py:26-40, groups=2)."""
    n, c, h, w = x.shape
    return x.view(n, 2, c // 2, h, w).transpose(1, 2).reshape(n, c, h, w)


def adaptive_avg_pool1d(x: torch.Tensor, m: int) -> torch.Tensor:
    """bin i = [floor(i*L/m), ceil((i+1)*L/m))"""
    L = x.shape[-1]
    cols = []
    for i in range(m):
        s = (i * L) // m
        e = -((-(i + 1) * L) // m)
        cols.append(x[..., s:e].mean(dim=-1))
    return torch.stack(cols, dim=-1)


# ----------------------------------------------------------------------------------
# visual encoder  (video.py:76-87)
# ----------------------------------------------------------------------------------
def _pw(x, sd, conv, bn, relu=True):
    y = batchnorm_eval(F.conv2d(x, sd[conv + ".weight"]), sd, bn)
    return torch.relu(y) if relu else y


def _dw(x, sd, conv, bn, stride):
    w = sd[conv + ".weight"]
    return batchnorm_eval(F.conv2d(x, w, stride=stride, padding=1, groups=w.shape[0]), sd, bn)


def shuffle_unit(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """One InvertedResidual (shufflenetv2.py:42-104).  A unit with a ``banch1`` is the
    stride-2 two-branch form, otherwise the stride-1 split form."""
    if (p + "banch1.0.weight") in sd:
        left = _pw(_dw(x, sd, p + "banch1.0", p + "banch1.1", 2), sd, p + "banch1.2", p + "banch1.3")
        r = _pw(x, sd, p + "banch2.0", p + "banch2.1")
        r = _dw(r, sd, p + "banch2.3", p + "banch2.4", 2)
        r = _pw(r, sd, p + "banch2.5", p + "banch2.6")
        out = torch.cat([left, r], dim=1)
    else:
        half = x.shape[1] // 2
        x1, x2 = x[:, :half], x[:, half:]
        r = _pw(x2, sd, p + "banch2.0", p + "banch2.1")
        r = _dw(r, sd, p + "banch2.3", p + "banch2.4", 1)
        r = _pw(r, sd, p + "banch2.5", p + "banch2.6")
        out = torch.cat([x1, r], dim=1)
    return channel_shuffle2(out)


def frontend3d(video: torch.Tensor, sd: SD, prefix: str = "encoder.") -> torch.Tensor:
    """(B,3,T,H,W) -> (B*T,24,H/4,W/4): Conv3d 5x7x7 s(1,2,2) p(2,3,3) + BN + PReLU + MaxPool(1,3,3)/s(1,2,2)/p(0,1,1),
    then frames folded into the batch, index b*T+t (video.py:20-23,68-72)."""
    p = prefix + "frontend3D."
    y = F.conv3d(video, sd[p + "0.weight"], stride=(1, 2, 2), padding=(2, 3, 3))
    y = batchnorm_eval(y, sd, p + "1")
    a = sd[p + "2.weight"].view(1, -1, 1, 1, 1)
    y = torch.where(y >= 0, y, a * y)
    y = F.max_pool3d(y, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    B, C, T, H, W = y.shape
    return y.transpose(1, 2).reshape(B * T, C, H, W)


def encoder_forward(sd: SD, video: torch.Tensor, prefix: str = "encoder.", taps: Optional[dict] = None) -> torch.Tensor:
    """(B,3,T,H,W) -> (B,T,768), unit L2 norm over the last dim."""
    B, _, T, _, _ = video.shape
    x = frontend3d(video, sd, prefix)
    if taps is not None:
        taps["frontend"] = x
    unit = 0
    while (f"{prefix}trunk.0.{unit}.banch2.0.weight") in sd:
        x = shuffle_unit(x, sd, f"{prefix}trunk.0.{unit}.")
        if taps is not None:
            taps[f"unit{unit}"] = x
        unit += 1
    x = _pw(x, sd, prefix + "trunk.1.0", prefix + "trunk.1.1")
    x = x.mean(dim=(2, 3))                       # AvgPool2d(3) on the 3x3 map
    x = x.view(B, T, -1)
    return x / torch.clamp(torch.sqrt((x * x).sum(dim=2, keepdim=True)), min=1e-12)


# ----------------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------------
def _conv1d_bn_silu(x, sd, p, stride=1, padding=0):
    y = F.conv1d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=stride, padding=padding)
    return silu(batchnorm_eval(y, sd, p + ".1"))


def multihop(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """MultiHopConv (decoder.py:159-196): cat[x, branch_k(x) for k in 1,3,7,11] -> 1x1 bottleneck. x:(B,512,T)."""
    feats = [x]
    for j, k in enumerate((1, 3, 7, 11)):
        feats.append(_conv1d_bn_silu(x, sd, f"{p}conv.{j}", padding=k // 2))
    return F.conv1d(torch.cat(feats, dim=1), sd[p + "bottleneck.weight"], sd[p + "bottleneck.bias"])


def content_encode(x: torch.Tensor, sd: SD, gumbel: torch.Tensor, p: str = "decoder.content."):
    """Content.encode (decoder.py:239-260).  x:(B,512,T).  ``gumbel`` (B*m,501) is the noise the reference
    draws inside F.gumbel_softmax - an explicit input here.  Returns key (B,256,m), value (B,m,256),
    content_dis (B*m,501)."""
    feats = [x]
    for j, k in enumerate((1, 3, 5, 7)):
        feats.append(_conv1d_bn_silu(x, sd, f"{p}agg.{j}", stride=k))
    m = min(f.shape[-1] for f in feats)
    cat = torch.cat([adaptive_avg_pool1d(f, m) for f in feats], dim=1)
    w = F.conv1d(cat, sd[p + "bottleneck.weight"], sd[p + "bottleneck.bias"]).permute(0, 2, 1)   # (B,m,256)
    key = silu(linear(silu(linear(w, sd, p + "K.0")), sd, p + "K.2")).permute(0, 2, 1)
    l = w
    for idx in (0, 2, 4):
        l = silu(linear(l, sd, f"{p}location_fc.{idx}"))
    B = x.shape[0]
    logits = l.reshape(B * m, -1)
    z = torch.softmax((logits + gumbel) / 0.1, dim=-1)
    value = (z @ sd[p + "word_embeddings"]).view(B, m, -1)
    return key, value, torch.softmax(logits, dim=-1)


def decoder_prologue(sd: SD, vis: torch.Tensor, emb: torch.Tensor, gumbel: torch.Tensor, p: str = "decoder."):
    """decoder.py:383-410.  vis (B,T,1024) = cat(visual features, tiled embedding); emb (B,256)."""
    B, T, _ = vis.shape
    residual = vis @ sd[p + "residual_bottleneck.weight"][:, :, 0].t() + sd[p + "residual_bottleneck.bias"]
    s_e = psine_last(linear(emb, sd, p + "encoder_site.0.linear_layer"), sd[p + "encoder_site.1.w"])
    s_a = psine_last(linear(emb, sd, p + "attention_site.0.linear_layer"), sd[p + "attention_site.1.w"])
    # BiLSTM, h0 = c0 = s_e for both directions
    outs = []
    finals = []
    for suf, order in (("l0", range(T)), ("l0_reverse", range(T - 1, -1, -1))):
        h, c = s_e, s_e
        seq = [None] * T
        wi, wh = sd[f"{p}encoder_rnn.weight_ih_{suf}"], sd[f"{p}encoder_rnn.weight_hh_{suf}"]
        bi, bh = sd[f"{p}encoder_rnn.bias_ih_{suf}"], sd[f"{p}encoder_rnn.bias_hh_{suf}"]
        for t in order:
            h, c = lstm_cell(vis[:, t], h, c, wi, wh, bi, bh)
            seq[t] = h
        outs.append(torch.stack(seq, dim=1))
        finals.append((h, c))
    rnn_out = torch.cat(outs, dim=2)                                   # (B,T,1024)
    hidden = torch.stack([finals[0][0], finals[1][0]], dim=0)          # (2,B,512)
    cell_cat = torch.cat([finals[0][1], finals[1][1]], dim=1)
    encoder_cell = linear(cell_cat, sd, p + "E_C.linear_layer")        # (B,512)
    enc = linear(rnn_out, sd, p + "encoder_proj.linear_layer") + s_a[:, None, :] + residual
    pos = sd[p + "positional_encodings.pos_table"][0, :T].t()          # (512,T)
    x = enc.permute(0, 2, 1)                                           # (B,512,T)
    k = psine_channels_first(multihop(x, sd, p + "K.0."), sd[p + "K.1.w"]) + pos
    v = (psine_channels_first(multihop(x, sd, p + "V.0."), sd[p + "V.1.w"]) + pos).permute(0, 2, 1)
    key, value, content_dis = content_encode(x, sd, gumbel, p + "content.")
    return dict(k=k, v=v, key=key, value=value, hidden=hidden, encoder_cell=encoder_cell,
                content_dis=content_dis, enc=enc)


def decode_loop(sd: SD, st: dict, S: int, p: str = "decoder.", teacher: Optional[torch.Tensor] = None,
                teacher_mask: Optional[torch.Tensor] = None, return_logits: bool = False, drop: Optional[dict] = None):
    """decoder.py:412-429 (inference) / 353-375 (forward).  ``teacher`` (B,S,80) with boolean ``teacher_mask`` (S,)
    substitutes the previous frame at the marked steps (scheduled sampling made explicit).  ``drop`` holds the train-mode dropout
    multipliers (0 or 1/(1-p)) as explicit tensors: 'prenet' (S,B,256) nn.Dropout(0.2) :308, 'attn' (S,B,T) F.dropout(logits, 0.1) :363
    (the returned logits are the dropped ones, as the reference appends after the dropout), 'rnn' (S,B,512) nn.LSTM(dropout=0.1) :312
    between the two layers.  Returns
    mel (B,S,80), stop logits (B,S), attention (B,S,T) (post-softmax, or tau*q.k logits if return_logits)."""
    k, v, key, value = st["k"], st["v"], st["key"], st["value"]
    h0, h1 = st["hidden"][0], st["hidden"][1]
    B = h0.shape[0]
    c0 = torch.zeros_like(h0)
    c1 = torch.zeros_like(h1)
    y = sd[p + "BOS"].view(1, -1).expand(B, -1)
    pos = sd[p + "positional_encodings.pos_table"][0]
    tau, tau_c = sd[p + "temperature"], sd[p + "content.temperature"]
    mels, stops, attns = [], [], []
    for i in range(S):
        if teacher is not None and teacher_mask is not None and bool(teacher_mask[i]):
            y = teacher[:, i]
        drop = drop or {}
        pr = psine_last(linear(y, sd, p + "prenet.0.linear_layer"), sd[p + "prenet.1.w"])
        if drop.get("prenet") is not None:
            pr = pr * drop["prenet"][i]
        pr = psine_last(linear(pr, sd, p + "prenet.3.linear_layer"), sd[p + "prenet.4.w"])
        q = psine_last(linear(torch.cat([h0, h1], dim=1), sd, p + "Q.0.linear_layer"), sd[p + "Q.1.w"]) + pos[i]
        logits = torch.bmm((q * tau).unsqueeze(1), k).squeeze(1)                  # (B,T)
        if drop.get("attn") is not None:
            logits = logits * drop["attn"][i]
        a = torch.softmax(logits, dim=-1)
        attns.append(logits if return_logits else a)
        o = linear(torch.bmm(a.unsqueeze(1), v).squeeze(1), sd, p + "attention_proj.linear_layer")
        u = pr + o
        qc = silu(linear(torch.cat([c0, c1], dim=1), sd, p + "content.Q.0"))
        al = torch.softmax(torch.bmm((qc * tau_c).unsqueeze(1), key).squeeze(1), dim=-1)
        cc = torch.bmm(al.unsqueeze(1), value).squeeze(1)
        h0, c0 = lstm_cell(torch.cat([cc, u], dim=1), h0, c0,
                           sd[p + "decoder_rnn.weight_ih_l0"], sd[p + "decoder_rnn.weight_hh_l0"],
                           sd[p + "decoder_rnn.bias_ih_l0"], sd[p + "decoder_rnn.bias_hh_l0"])
        h1, c1 = lstm_cell(h0 if drop.get("rnn") is None else h0 * drop["rnn"][i], h1, c1,
                           sd[p + "decoder_rnn.weight_ih_l1"], sd[p + "decoder_rnn.weight_hh_l1"],
                           sd[p + "decoder_rnn.bias_ih_l1"], sd[p + "decoder_rnn.bias_hh_l1"])
        y = linear(h1, sd, p + "fc_out.linear_layer")
        mels.append(y)
        stops.append(linear(torch.cat([h1, st["encoder_cell"]], dim=1), sd, p + "stop_token_layer.linear_layer")[:, 0])
    return torch.stack(mels, dim=1), torch.stack(stops, dim=1), torch.stack(attns, dim=1)


def output_lengths_from_stop(stop_logits: torch.Tensor, S: int) -> torch.Tensor:
    """first i+1 with sigmoid(stop)>0.5 (<=> logit>0), else S; int64 (decoder.py:429-435)."""
    hit = stop_logits > 0
    first = torch.where(hit.any(dim=1), hit.float().argmax(dim=1) + 1, torch.full((stop_logits.shape[0],), S))
    return first.to(torch.int64)


def postnet(sd: SD, mel: torch.Tensor, p: str = "decoder.postnet.", drop=None) -> torch.Tensor:
    """decoder.py:143-156: mel (B,80,S) -> residual correction (B,80,S) (caller adds mel).  ``drop``: the five train-mode dropout
    multipliers (p 0.5, :152,154) as explicit tensors (B,C,S), applied after the residual add / after the last conv."""
    x = mel
    for i in range(5):
        y = F.conv1d(x, sd[f"{p}convolutions.{i}.0.conv.weight"], sd[f"{p}convolutions.{i}.0.conv.bias"], padding=2)
        y = batchnorm_eval(y, sd, f"{p}convolutions.{i}.1")
        if i < 4:
            y = psine_channels_first(y, sd[f"{p}sin_activation.{i}.w"])
            if i != 0:
                y = y + x
        if drop is not None:
            y = y * drop[i]
        x = y
    return x


def build_visual(feat: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """model.py:52-55: tile the (B,256) embedding over T and concatenate -> (B,T,1024)."""
    return torch.cat([feat, emb[:, None, :].expand(-1, feat.shape[1], -1)], dim=2)


def inference(sd: SD, video: torch.Tensor, emb: torch.Tensor, gumbel: torch.Tensor, S: int = 300, taps: Optional[dict] = None):
    """Lip2Speech.inference with a supplied speaker embedding (model.py:43-59).
    Returns mel_post (B,80,S), output_lengths (B,) int64, attention (B,S,T), and extras."""
    feat = encoder_forward(sd, video)
    st = decoder_prologue(sd, build_visual(feat, emb), emb, gumbel)
    mel, stop, attn = decode_loop(sd, st, S)
    mel_cf = mel.permute(0, 2, 1)
    mel_post = postnet(sd, mel_cf) + mel_cf
    if taps is not None:
        taps.update(feat=feat, mel=mel_cf, stop=stop, **st)
    return mel_post, output_lengths_from_stop(stop, S), attn


def forward_eval(sd: SD, video: torch.Tensor, emb: torch.Tensor, mels: torch.Tensor, gumbel: torch.Tensor,
                 teacher_mask: Optional[torch.Tensor] = None):
    """Lip2Speech.forward in eval() with tf_ratio=1 semantics (evaluate.py:38): S = mels.shape[2], free running
    unless ``teacher_mask`` marks steps; returns the reference's list
    [mel (B,80,S), mel_post, stop (B,S,1), emb (B,256), attention LOGITS (B,S,T), content_dis]."""
    feat = encoder_forward(sd, video)
    st = decoder_prologue(sd, build_visual(feat, emb), emb, gumbel)
    S = mels.shape[2]
    bos = sd["decoder.BOS"].view(1, 1, -1).expand(mels.shape[0], -1, -1)
    teacher = torch.cat([bos, mels.permute(0, 2, 1)], dim=1)
    mel, stop, attn = decode_loop(sd, st, S, teacher=teacher, teacher_mask=teacher_mask, return_logits=True)
    mel_cf = mel.permute(0, 2, 1)
    return [mel_cf, postnet(sd, mel_cf) + mel_cf, stop.unsqueeze(2), emb, attn, st["content_dis"]]


def loss_terms(outs, mel_target: torch.Tensor, gate_target: torch.Tensor):
    """Loss.forward (train_utils/losses.py:69-77) + train.py:175: returns (mel_loss, postnet_mel_loss, gate_loss, KLD, total)."""
    mel, mel_post, stop, qy = outs[0], outs[1], outs[2], outs[5]
    kld = torch.sum(qy * torch.log(qy * qy.shape[-1] + 1e-20), dim=-1).mean()
    mel_loss = F.mse_loss(mel, mel_target)
    post_loss = 10 * F.mse_loss(mel_post, mel_target)
    gate_loss = F.binary_cross_entropy_with_logits(stop.reshape(-1, 1), gate_target.reshape(-1, 1))
    return mel_loss, post_loss, gate_loss, kld, kld + mel_loss + post_loss + gate_loss


# ----------------------------------------------------------------------------------
# speaker encoder  (audio.py:110-150).  The mel front-end is torchaudio's (absent here): its published algorithm is
# restated below - PARITY UNPINNED for that piece; the LSTM/Linear tail is plain nn.LSTM math.
# ----------------------------------------------------------------------------------
def htk_mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int, dtype=torch.float64) -> torch.Tensor:
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=dtype)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10.0 ** (torch.linspace(m_min, m_max, n_mels + 2, dtype=dtype) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)


def mel40(audio: torch.Tensor) -> torch.Tensor:
    """(B,N) -> (B,L,40): power spectrogram n_fft 400 / hop 160 / hann / centre-reflect, HTK mel 0-8 kHz, no log."""
    win = torch.hann_window(400, periodic=True, dtype=audio.dtype)
    spec = torch.stft(audio, 400, hop_length=160, win_length=400, window=win, center=True, pad_mode="reflect", return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    fb = htk_mel_filterbank(201, 0.0, 8000.0, 40, 16000).to(audio.dtype)
    return power.transpose(1, 2) @ fb


def speaker_lstm_tail(sd: SD, mel: torch.Tensor, p: str = "speaker_encoder.") -> torch.Tensor:
    """3 x LSTM(256) from a zero state over (B,L,40) -> Linear(h_last) -> ReLU -> L2 normalise."""
    x = mel
    B, L, _ = x.shape
    for layer in range(3):
        h = torch.zeros(B, 256, dtype=x.dtype)
        c = torch.zeros(B, 256, dtype=x.dtype)
        outs = []
        for t in range(L):
            h, c = lstm_cell(x[:, t], h, c, sd[f"{p}lstm.weight_ih_l{layer}"], sd[f"{p}lstm.weight_hh_l{layer}"],
                             sd[f"{p}lstm.bias_ih_l{layer}"], sd[f"{p}lstm.bias_hh_l{layer}"])
            outs.append(h)
        x = torch.stack(outs, dim=1)
    e = torch.relu(x[:, -1] @ sd[p + "linear.weight"].t() + sd[p + "linear.bias"])
    return e / torch.clamp(torch.sqrt((e * e).sum(dim=1, keepdim=True)), min=1e-12)


def speaker_encoder_inference(sd: SD, audio: torch.Tensor) -> torch.Tensor:
    return speaker_lstm_tail(sd, mel40(audio))


def to_dtype(sd: SD, dtype) -> SD:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
