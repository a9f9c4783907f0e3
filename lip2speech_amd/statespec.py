"""Checkpoint key table for the hot path (the on-disk format of the boundary).

The reference stores its weights as a ``torch.save``d ``state_dict`` whose key
names are the contract; ``demo.py:30-38`` loads them with
``strict=True``.  This module states that table once, as data:

    (key, shape, kind)

``kind`` tells (a) whether the entry is a parameter or a buffer and (b) which
initial-value family ``synth.py`` draws it from.  The key names - including the
reference's ``banch1``/``banch2`` spelling - follow

  * encoder.*  : /root/reference/model/modules/video.py:55-72 and
                 /root/reference/model/modules/shufflenetv2.py:42-152
  * decoder.*  : /root/reference/model/modules/decoder.py:107-318

Nothing here is executable model code; the arithmetic lives in csrc/ (HIP) and,
for checking only, in oracle/.
"""
from __future__ import annotations

from typing import List, Tuple

Spec = List[Tuple[str, Tuple[int, ...], str]]

# ShuffleNetV2 1.0x geometry used by the visual encoder
STAGE_CH = (24, 116, 232, 464)          # channels entering stage 2 and leaving stages 2..4
STAGE_REPEATS = (4, 8, 4)
LAST_CH = 768
FRONT_CH = 24

# decoder geometry (reference/hparams.py + decoder.py:285-318)
N_MELS = 80
D_MODEL = 512
D_ENC = 1024
D_EMB = 256
D_PRENET = 256
D_CONTENT = 256
VOCAB = 501
MAX_STEPS = 300
MULTIHOP_KS = (1, 3, 7, 11)
CONTENT_KS = (1, 3, 5, 7)
POSTNET_K = 5
POSTNET_LAYERS = 5

BUFFER_KINDS = {"bn_rm", "bn_rv", "bn_nbt", "pos_table"}


def _bn(prefix: str, c: int) -> Spec:
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_rm"),
        (prefix + ".running_var", (c,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def encoder_spec(prefix: str = "") -> Spec:
    s: Spec = []
    # trunk.0 = the 16 ShuffleNet units, trunk.1 = conv_last (+BN), trunk.2 = avg-pool (no state)
    unit = 0
    cin = STAGE_CH[0]
    for stage, rep in enumerate(STAGE_REPEATS):
        cout = STAGE_CH[stage + 1]
        half = cout // 2
        for r in range(rep):
            p = f"{prefix}trunk.0.{unit}."
            if r == 0:  # stride-2 unit: two branches over the whole input
                s.append((p + "banch1.0.weight", (cin, 1, 3, 3), "conv_enc"))
                s += _bn(p + "banch1.1", cin)
                s.append((p + "banch1.2.weight", (half, cin, 1, 1), "conv_enc"))
                s += _bn(p + "banch1.3", half)
                s.append((p + "banch2.0.weight", (half, cin, 1, 1), "conv_enc"))
            else:       # stride-1 unit: one branch over the second channel half
                s.append((p + "banch2.0.weight", (half, half, 1, 1), "conv_enc"))
            s += _bn(p + "banch2.1", half)
            s.append((p + "banch2.3.weight", (half, 1, 3, 3), "conv_enc"))
            s += _bn(p + "banch2.4", half)
            s.append((p + "banch2.5.weight", (half, half, 1, 1), "conv_enc"))
            s += _bn(p + "banch2.6", half)
            cin = cout
            unit += 1
    s.append((f"{prefix}trunk.1.0.weight", (LAST_CH, STAGE_CH[-1], 1, 1), "conv_enc"))
    s += _bn(f"{prefix}trunk.1.1", LAST_CH)
    s.append((f"{prefix}frontend3D.0.weight", (FRONT_CH, 3, 5, 7, 7), "conv_enc"))
    s += _bn(f"{prefix}frontend3D.1", FRONT_CH)
    s.append((f"{prefix}frontend3D.2.weight", (FRONT_CH,), "prelu"))
    return s


def _linear(prefix: str, cout: int, cin: int, gain: str) -> Spec:
    return [(prefix + ".weight", (cout, cin), "xavier:" + gain),
            (prefix + ".bias", (cout,), f"bias:{cin}")]


def _conv1d(prefix: str, cout: int, cin: int, k: int, kind: str) -> Spec:
    return [(prefix + ".weight", (cout, cin, k), kind),
            (prefix + ".bias", (cout,), f"bias:{cin * k}")]


def _lstm(prefix: str, inp: int, hid: int, suffixes) -> Spec:
    s: Spec = []
    for suf in suffixes:
        s.append((f"{prefix}.weight_ih_{suf}", (4 * hid, inp), f"lstm:{hid}"))
        s.append((f"{prefix}.weight_hh_{suf}", (4 * hid, hid), f"lstm:{hid}"))
        s.append((f"{prefix}.bias_ih_{suf}", (4 * hid,), f"lstm:{hid}"))
        s.append((f"{prefix}.bias_hh_{suf}", (4 * hid,), f"lstm:{hid}"))
    return s


def _multihop(prefix: str, ks, cout: int, strided: bool) -> Spec:
    """MultiHopConv (decoder.py:159-196) / Content.agg (decoder.py:208-236) share a shape."""
    s: Spec = []
    name = "agg" if strided else "conv"
    for j, k in enumerate(ks):
        s += _conv1d(f"{prefix}{name}.{j}.0", D_MODEL, D_MODEL, k, "default")
        s += _bn(f"{prefix}{name}.{j}.1", D_MODEL)
    s += _conv1d(f"{prefix}bottleneck", cout, D_MODEL * (len(ks) + 1), 1, "default")
    return s


def decoder_spec(prefix: str = "") -> Spec:
    p = prefix
    s: Spec = [(p + "BOS", (1, 1, N_MELS), "bos"), (p + "temperature", (1,), f"temp:{D_MODEL}")]
    for i in range(POSTNET_LAYERS):
        cin = N_MELS if i == 0 else D_MODEL
        cout = N_MELS if i == POSTNET_LAYERS - 1 else D_MODEL
        gain = "linear" if i == POSTNET_LAYERS - 1 else "tanh"
        s += _conv1d(f"{p}postnet.convolutions.{i}.0.conv", cout, cin, POSTNET_K, "xavier:" + gain)
        s += _bn(f"{p}postnet.convolutions.{i}.1", cout)
    for i in range(POSTNET_LAYERS - 1):
        s.append((f"{p}postnet.sin_activation.{i}.w", (D_MODEL,), "psine"))
    s += _linear(p + "encoder_proj.linear_layer", D_MODEL, 2 * D_MODEL, "linear")
    for site in ("encoder_site", "attention_site"):
        s += _linear(f"{p}{site}.0.linear_layer", D_MODEL, D_EMB, "linear")
        s.append((f"{p}{site}.1.w", (D_MODEL,), "psine"))
    s += _conv1d(p + "residual_bottleneck", D_MODEL, D_ENC, 1, "default")
    s += _lstm(p + "encoder_rnn", D_ENC, D_MODEL, ("l0", "l0_reverse"))
    for kv in ("K", "V"):
        s += _multihop(f"{p}{kv}.0.", MULTIHOP_KS, D_MODEL, strided=False)
        s.append((f"{p}{kv}.1.w", (D_MODEL,), "psine"))
    s += _linear(p + "Q.0.linear_layer", D_MODEL, 2 * D_MODEL, "linear")
    s.append((p + "Q.1.w", (D_MODEL,), "psine"))
    s.append((p + "content.word_embeddings", (VOCAB, D_CONTENT), "emb"))
    s.append((p + "content.temperature", (1,), f"temp:{D_CONTENT}"))
    s += _multihop(p + "content.", CONTENT_KS, D_CONTENT, strided=True)
    for idx, (co, ci) in zip((0, 2, 4), ((D_CONTENT, D_CONTENT), (D_CONTENT, D_CONTENT), (VOCAB, D_CONTENT))):
        s += [(f"{p}content.location_fc.{idx}.weight", (co, ci), "default"),
              (f"{p}content.location_fc.{idx}.bias", (co,), f"bias:{ci}")]
    for idx in (0, 2):
        s += [(f"{p}content.K.{idx}.weight", (D_CONTENT, D_CONTENT), "default"),
              (f"{p}content.K.{idx}.bias", (D_CONTENT,), f"bias:{D_CONTENT}")]
    s += [(p + "content.Q.0.weight", (D_CONTENT, 2 * D_MODEL), "default"),
          (p + "content.Q.0.bias", (D_CONTENT,), f"bias:{2 * D_MODEL}")]
    s += _linear(p + "attention_proj.linear_layer", D_PRENET, D_MODEL, "linear")
    s += _linear(p + "prenet.0.linear_layer", D_PRENET, N_MELS, "linear")
    s.append((p + "prenet.1.w", (D_PRENET,), "psine"))
    s += _linear(p + "prenet.3.linear_layer", D_PRENET, D_PRENET, "linear")
    s.append((p + "prenet.4.w", (D_PRENET,), "psine"))
    s += _lstm(p + "decoder_rnn", D_MODEL, D_MODEL, ("l0", "l1"))
    s += _linear(p + "fc_out.linear_layer", N_MELS, D_MODEL, "linear")
    s += _linear(p + "E_C.linear_layer", D_MODEL, 2 * D_MODEL, "sigmoid")
    s += _linear(p + "stop_token_layer.linear_layer", 1, 2 * D_MODEL, "sigmoid")
    s.append((p + "positional_encodings.pos_table", (1, MAX_STEPS, D_MODEL), "pos_table"))
    return s


def speaker_encoder_spec(prefix: str = "") -> Spec:
    """speaker_encoder.* keys (reference/model/modules/audio.py:110-121)."""
    s: Spec = []
    for layer in range(3):
        inp = 40 if layer == 0 else 256
        s.append((f"{prefix}lstm.weight_ih_l{layer}", (1024, inp), "lstm:256"))
        s.append((f"{prefix}lstm.weight_hh_l{layer}", (1024, 256), "lstm:256"))
        s.append((f"{prefix}lstm.bias_ih_l{layer}", (1024,), "lstm:256"))
        s.append((f"{prefix}lstm.bias_hh_l{layer}", (1024,), "lstm:256"))
    s += [(prefix + "linear.weight", (256, 256), "default"), (prefix + "linear.bias", (256,), "bias:256")]
    return s


def model_spec() -> Spec:
    """Keys of the measured path inside a Lip2Speech checkpoint (vgg_face.* is the
    third-party face tower, outside this path - SURVEY.md §2 row 6)."""
    return encoder_spec("encoder.") + decoder_spec("decoder.")
