#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  It imports the
reference's hot-path modules by file path (they need only torch + numpy), loads the
repo's deterministic synthetic checkpoint (lip2speech_amd.synth) into them, feeds the
Gumbel noise explicitly (the reference draws it inside F.gumbel_softmax even in eval,
decoder.py:257) and stores inputs that cannot be regenerated bit-stably (the noise)
plus the reference outputs.  Videos / embeddings / weights are NOT stored: they are
regenerated from integers by lip2speech_amd.synth on every host.

It also cross-checks oracle/l2s_oracle.py against the reference and prints the
deviations, so a failing oracle never produces "goldens".

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from lip2speech_amd import statespec, synth          # noqa: E402
from oracle import l2s_oracle as orc                  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)                                       # for `hparams`
    _load("shufflenetv2", f"{REF}/model/modules/shufflenetv2.py")  # video.py tries the absolute import first
    vid = _load("ref_video", f"{REF}/model/modules/video.py")
    dec = _load("ref_decoder", f"{REF}/model/modules/decoder.py")
    return vid, dec


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


class GumbelFeed:
    """Replace F.gumbel_softmax inside the reference decoder by softmax((logits+G)/tau) with a supplied G."""

    def __init__(self, dec_mod, noise):
        self.mod, self.noise = dec_mod, noise

    def __enter__(self):
        self.orig = self.mod.F.gumbel_softmax
        noise = self.noise

        def fed(logits, tau=1, hard=False, eps=1e-10, dim=-1):
            assert logits.shape == noise.shape, (logits.shape, noise.shape)
            return ((logits + noise.to(logits.dtype)) / tau).softmax(dim)

        self.mod.F.gumbel_softmax = fed

    def __exit__(self, *a):
        self.mod.F.gumbel_softmax = self.orig


def min_T(T):
    return min(T, (T - 3) // 3 + 1, (T - 5) // 5 + 1, (T - 7) // 7 + 1)


def top2(a):
    """argmax over the last dim and the margin between the two largest entries."""
    srt, idx = torch.sort(a, dim=-1, descending=True)
    return idx[..., 0].to(torch.int32), (srt[..., 0] - srt[..., 1])


def report(tag, got, want):
    """max|d|; attention LOGITS are tau*q.k with |values| in the hundreds, so they are compared relative to their scale."""
    d = (got.double() - want.double()).abs().max().item()
    scale = max(1.0, want.double().abs().max().item()) if "logits" in tag else 1.0
    print(f"   oracle-vs-reference {tag:<22s} max|d| = {d:.3e}" + (f"  (scale {scale:.1f})" if scale != 1.0 else ""))
    return d / scale


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vid, dec = load_reference()
    sd = synth.synth_state_dict()

    enc_ref = vid.VideoExtractor().eval()
    dec_ref = dec.Decoder().eval()
    # key / shape contract of the boundary
    for mod, prefix, spec in ((enc_ref, "encoder.", statespec.encoder_spec("encoder.")),
                              (dec_ref, "decoder.", statespec.decoder_spec("decoder."))):
        ref_keys = {prefix + k: tuple(v.shape) for k, v in mod.state_dict().items()}
        my_keys = {k: tuple(s) for k, s, _ in spec}
        assert ref_keys == my_keys, set(ref_keys.items()) ^ set(my_keys.items())
    enc_ref.load_state_dict(sub(sd, "encoder."), strict=True)
    dec_ref.load_state_dict(sub(sd, "decoder."), strict=True)
    # the pos_table we regenerate must equal the reference's own buffer bit for bit
    fresh = dec.Decoder().state_dict()["positional_encodings.pos_table"]
    assert torch.equal(fresh, sd["decoder.positional_encodings.pos_table"]), "pos_table differs from reference"
    print("state_dict keys/shapes match the reference; pos_table bit-identical")

    worst = 0.0

    # ---------------- case 1: LRW-shaped inference, B=2 (config 1 shape) ----------------
    B, T, S = 2, 29, 300
    video = synth.synth_video(B, T, tag="video-lrw2")
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2")
    gum = synth.synth_gumbel(B * min_T(T), tag="gumbel-lrw2")
    with torch.no_grad(), GumbelFeed(dec, gum):
        feat = enc_ref(video)
        face = emb.unsqueeze(1).repeat(1, T, 1)
        vis = torch.cat([feat, face], dim=2)
        mel_post, lengths, attn = dec_ref.inference(vis, face, return_attention_map=True)
    taps = {}
    with torch.no_grad():
        o_post, o_len, o_attn = orc.inference(sd, video, emb, gum, S=S, taps=taps)
        enc_taps = {}
        orc.encoder_forward(sd, video, taps=enc_taps)
    worst = max(worst, report("feat", taps["feat"], feat))
    worst = max(worst, report("mel_post(S=300)", o_post, mel_post))
    worst = max(worst, report("attention", o_attn, attn))
    assert torch.equal(o_len, lengths), (o_len, lengths)
    # the reference does not return the pre-postnet mel from inference(); recover it exactly: postnet is
    # deterministic, so mel_pre = the oracle's, verified through mel_post above and through forward() below.
    amax, margin = top2(attn)
    o_amax, _ = top2(o_attn)
    print("   attention argmax equal:", bool((amax == o_amax).all()), " min top-2 margin:", margin.min().item())
    frames = [0, T + 1]
    np.savez_compressed(
        os.path.join(HERE, "inference_lrw_b2.npz"),
        gumbel=gum.numpy(), feat=feat.numpy(), mel_post=mel_post.numpy(), output_lengths=lengths.numpy(),
        attn_argmax=amax.numpy(), attn_margin=margin.numpy().astype(np.float32),
        attn_rows=attn[:, ::50].numpy(),
        oracle_mel_pre=taps["mel"].numpy(), oracle_stop=taps["stop"].numpy(),
        oracle_k=taps["k"].numpy(), oracle_v=taps["v"].numpy(), oracle_key=taps["key"].numpy(),
        oracle_value=taps["value"].numpy(), oracle_hidden=taps["hidden"].numpy(),
        oracle_encoder_cell=taps["encoder_cell"].numpy(), oracle_enc=taps["enc"].numpy(),
        frames=np.asarray(frames),
        oracle_frontend=enc_taps["frontend"][frames].numpy(), oracle_unit0=enc_taps["unit0"][frames].numpy(),
        oracle_unit3=enc_taps["unit3"][frames].numpy(), oracle_unit4=enc_taps["unit4"][frames].numpy(),
        oracle_unit11=enc_taps["unit11"][frames].numpy(), oracle_unit15=enc_taps["unit15"][frames].numpy(),
    )
    # stage checkpoints of the reference encoder itself (frontend3D + trunk pieces) to pin the oracle's taps
    with torch.no_grad():
        fr = vid.threeD_to_2D_tensor(enc_ref.frontend3D(video))
        worst = max(worst, report("frontend3D", enc_taps["frontend"], fr))
        x = fr
        for ui, unit in enumerate(enc_ref.trunk[0]):
            x = unit(x)
            if ui in (0, 3, 4, 11, 15):
                worst = max(worst, report(f"unit{ui}", enc_taps[f"unit{ui}"], x))

    # ---------------- case 2: evaluate.py semantics: forward(tf_ratio=1), S=77 ----------------
    S = 77
    mels = synth.synth_mels(B, S, tag="mel-lrw2")
    with torch.no_grad(), GumbelFeed(dec, gum):
        outs = dec_ref(vis, face, mels, torch.full((B,), T), torch.full((B,), S), 1)
        o = orc.forward_eval(sd, video, emb, mels, gum)
    names = ["mel", "mel_post", "stop", "emb", "attn_logits", "content_dis"]
    for n, a, b in zip(names, o, outs):
        worst = max(worst, report(f"forward.{n}", a, b))
    np.savez_compressed(os.path.join(HERE, "forward_lrw_b2_s77.npz"),
                        **{n: t.numpy() for n, t in zip(names, outs)})
    # pre-postnet mel of inference() == first 77 steps of forward(tf=1) (same recurrence): pins oracle_mel_pre
    worst = max(worst, report("mel_pre[:77] inf-vs-fwd", taps["mel"][:, :, :77], outs[0]))

    # ---------------- case 3: GRID-like variable T (T=75 -> min_T=10, S=188) ----------------
    B, T, S = 2, 75, 188
    video3 = synth.synth_video(B, T, tag="video-grid2")
    emb3 = synth.synth_speaker_embedding(B, tag="spk-grid2")
    gum3 = synth.synth_gumbel(B * min_T(T), tag="gumbel-grid2")
    mels3 = synth.synth_mels(B, S, tag="mel-grid2")
    with torch.no_grad(), GumbelFeed(dec, gum3):
        feat3 = enc_ref(video3)
        face3 = emb3.unsqueeze(1).repeat(1, T, 1)
        outs3 = dec_ref(torch.cat([feat3, face3], dim=2), face3, mels3, torch.full((B,), T), torch.full((B,), S), 1)
        o3 = orc.forward_eval(sd, video3, emb3, mels3, gum3)
    for n, a, b in zip(names, o3, outs3):
        worst = max(worst, report(f"grid.{n}", a, b))
    np.savez_compressed(os.path.join(HERE, "forward_grid_b2_t75_s188.npz"), gumbel=gum3.numpy(),
                        feat=feat3.numpy(), **{n: t.numpy() for n, t in zip(names, outs3)})

    # ---------------- case 4: padded batch, lengths (25,50) zero-padded to 50: lengths are ignored ------------
    B, T, S = 2, 50, 128
    video4 = synth.synth_video(B, T, tag="video-pad2")
    video4[0, :, 25:] = 0          # train_collate_fn_pad pads the shorter clip with zero frames
    emb4 = synth.synth_speaker_embedding(B, tag="spk-pad2")
    gum4 = synth.synth_gumbel(B * min_T(T), tag="gumbel-pad2")
    mels4 = synth.synth_mels(B, S, tag="mel-pad2")
    with torch.no_grad(), GumbelFeed(dec, gum4):
        feat4 = enc_ref(video4)
        face4 = emb4.unsqueeze(1).repeat(1, T, 1)
        outs4 = dec_ref(torch.cat([feat4, face4], dim=2), face4, mels4, torch.tensor([25, 50]), torch.full((B,), S), 1)
        o4 = orc.forward_eval(sd, video4, emb4, mels4, gum4)
    for n, a, b in zip(names, o4, outs4):
        worst = max(worst, report(f"pad.{n}", a, b))
    np.savez_compressed(os.path.join(HERE, "forward_pad_b2_t50_s128.npz"), gumbel=gum4.numpy(),
                        feat=feat4.numpy(), **{n: t.numpy() for n, t in zip(names, outs4)})

    # ---------------- case 5: teacher-forced steps (scheduled sampling made explicit) ----------------
    # The reference decides per step with torch.rand(1) > tf_ratio (decoder.py:355); reproduce its draws.
    B, T, S, tf = 2, 29, 77, 0.5
    torch.manual_seed(4321)
    draws = torch.stack([torch.rand(1) for _ in range(S)]).view(-1)
    mask = torch.zeros(S, dtype=torch.bool)
    consumed = 0
    for i in range(S):
        if draws[i] > tf and consumed < int(tf * S):
            consumed += 1
            mask[i] = True
    torch.manual_seed(4321)
    with torch.no_grad(), GumbelFeed(dec, gum):
        outs5 = dec_ref(vis, face, mels, torch.full((B,), T), torch.full((B,), S), tf)
        o5 = orc.forward_eval(sd, video, emb, mels, gum, teacher_mask=mask)
    for n, a, b in zip(names, o5, outs5):
        worst = max(worst, report(f"tf0.5.{n}", a, b))
    np.savez_compressed(os.path.join(HERE, "forward_lrw_b2_s77_tf05.npz"), teacher_mask=mask.numpy(),
                        **{n: t.numpy() for n, t in zip(names, outs5)})

    # ---------------- noise floor of the arithmetic itself: fp64 oracle vs fp32 reference ----------------
    sd64 = orc.to_dtype(sd, torch.float64)
    with torch.no_grad():
        p64, l64, a64 = orc.inference(sd64, video.double(), emb.double(), gum.double(), S=300)
    print(f"   fp64-oracle vs fp32-reference mel_post max|d| = {(p64 - mel_post.double()).abs().max().item():.3e}"
          f"  argmax equal: {bool((top2(a64)[0] == amax).all())}")
    print(f"worst oracle-vs-reference deviation: {worst:.3e}")
    assert worst < 5e-4, "oracle does not reproduce the reference"


if __name__ == "__main__":
    main()
