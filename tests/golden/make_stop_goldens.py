#!/usr/bin/env python
"""Golden vectors of THE REFERENCE for the stop bookkeeping (decoder.py:429-435) with NON-TRIVIAL stop behaviour.

With the synthetic checkpoint's random-init stop layer every clip of every other fixture stops at step 1-3.  Here the checkpoint
differs ONLY in `decoder.stop_token_layer.linear_layer.{weight,bias}`: a seeded random direction (the encoder_cell half scaled by
1/4) and a bias at a quantile of the resulting logits, picked by a deterministic search over (seed, quantile) so that the
reference's first crossings spread over the 300 steps, several clips never stop (-> 300), and every logit up to and including a
clip's first crossing is further from zero than MARGIN x std(logits) - a decision no correct fp32 implementation can flip.
The stop layer does not feed back into the recurrence, so mel and attention are those of the existing fixtures; stored here are
the chosen weight / bias, the reference's `output_lengths` (int64) and its stop logits (captured by a forward hook on the
reference's own `stop_token_layer`), for

  * `stop_lrw_b32.npz`  bench.py's batch (B=32, T=29, S=300; noise = inference_lrw_b32_full.npz's);
  * `stop_lrw_b2.npz`   BASELINE config 1's shape (B=2): one clip stops mid-sequence, the other never; plus the FULL (2,300,29)
                        post-softmax attention tensor of the reference (inference_lrw_b2.npz keeps every 50th row only).

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_stop_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg          # noqa: E402  (load_reference, GumbelFeed, sub, min_T)

from lip2speech_amd import synth   # noqa: E402
from oracle import l2s_oracle as orc   # noqa: E402

MARGIN = 8e-3
WKEY = "decoder.stop_token_layer.linear_layer.weight"
BKEY = "decoder.stop_token_layer.linear_layer.bias"


def first_crossings(s, S):
    """decoder.py:429-435 on a (B,S) logit table: first i+1 with sigmoid(stop) > 0.5 (<=> logit > 0), else S."""
    pos = s > 0
    return torch.where(pos.any(1), pos.float().argmax(1) + 1, torch.full((s.shape[0],), S)).to(torch.int64)


def run_reference(enc_ref, dec_ref, dec, video, emb, gum, T):
    """Reference inference; returns (mel_post, lengths, attn, stop-layer inputs (B,S,1024), stop logits (B,S))."""
    B = video.shape[0]
    ins, outs = [], []

    def hook(mod, i, o):
        ins.append(i[0].detach().clone().reshape(B, -1))
        outs.append(o.detach().clone().reshape(B))

    h = dec_ref.stop_token_layer.register_forward_hook(hook)
    try:
        with torch.no_grad(), mg.GumbelFeed(dec, gum):
            feat = enc_ref(video)
            face = emb.unsqueeze(1).repeat(1, T, 1)
            mel_post, lengths, attn = dec_ref.inference(torch.cat([feat, face], dim=2), face, return_attention_map=True)
    finally:
        h.remove()
    return mel_post, lengths, attn, torch.stack(ins, 1), torch.stack(outs, 1)


def search(X, accept, seeds=2000):
    """Deterministic search of (weight, bias): seeded N(0,1) direction, encoder_cell half x 1/4, bias = -quantile of the logits."""
    S = X.shape[1]
    best = None
    for seed in range(seeds):
        g = torch.Generator().manual_seed(seed)
        w = torch.randn(1024, generator=g)
        w[512:] *= 0.25
        s = X @ w
        for q in (0.6, 0.7, 0.8, 0.9):
            b = -torch.quantile(s.flatten(), q)
            ln = first_crossings(s + b, S)
            m = min((s[i, :ln[i]] + b).abs().min().item() for i in range(X.shape[0])) / s.std().item()
            score = accept(ln, m)
            if score is not None and (best is None or score > best[0]):
                best = (score, seed, q, w.clone(), b.clone(), ln)
    assert best is not None, "no stop layer found"
    return best


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vid, dec = mg.load_reference()
    sd = synth.synth_state_dict()
    enc_ref = vid.VideoExtractor().eval()
    dec_ref = dec.Decoder().eval()
    enc_ref.load_state_dict(mg.sub(sd, "encoder."), strict=True)
    dec_ref.load_state_dict(mg.sub(sd, "decoder."), strict=True)

    def case(name, B, T, video, emb, gum, accept, extra=None):
        S = 300
        dec_ref.load_state_dict(mg.sub(sd, "decoder."), strict=True)
        mel0, len0, attn0, X, st0 = run_reference(enc_ref, dec_ref, dec, video, emb, gum, T)
        print(f"{name}: stock stop layer -> lengths {sorted(len0.tolist())}")
        score, seed, q, w, b, ln = search(X, accept)
        print(f"{name}: picked seed {seed}, quantile {q}, score {score}: lengths {sorted(ln.tolist())}")
        sd2 = dict(sd)
        sd2[WKEY] = w.view(1, 1024).contiguous()
        sd2[BKEY] = b.view(1).contiguous()
        assert sd2[WKEY].shape == sd[WKEY].shape and sd2[BKEY].shape == sd[BKEY].shape
        dec_ref.load_state_dict(mg.sub(sd2, "decoder."), strict=True)
        mel, lengths, attn, X2, st = run_reference(enc_ref, dec_ref, dec, video, emb, gum, T)
        assert torch.equal(mel, mel0) and torch.equal(attn, attn0) and torch.equal(X, X2), "the stop layer fed back into the recurrence?"
        assert lengths.dtype == torch.int64 and torch.equal(lengths, first_crossings(st, S)) and torch.equal(lengths, ln)
        with torch.no_grad():
            taps = {}
            o_post, o_len, o_attn = orc.inference(sd2, video, emb, gum, S=S, taps=taps)
        print(f"   oracle-vs-reference: lengths equal {bool(torch.equal(o_len, lengths))}, stop logits max|d| = "
              f"{(taps['stop'].reshape(B, S) - st).abs().max().item():.3e} (scale {st.abs().max().item():.2f}), mel_post max|d| = {(o_post - mel).abs().max().item():.3e}")
        assert torch.equal(o_len, lengths)
        out = dict(stop_weight=sd2[WKEY].numpy(), stop_bias=sd2[BKEY].numpy(), output_lengths=lengths.numpy(), stop_logits=st.numpy(),
                   gumbel=gum.numpy())
        if extra:
            out.update(extra(mel, attn))
        np.savez_compressed(os.path.join(HERE, name), **out)
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)) // 1024, "KiB")

    # ---- B=32: bench.py's batch.  Want: >= 5 distinct lengths in (10, 300), >= 2 clips at 300, some early stops too.
    def accept32(ln, m):
        vals = ln.tolist()
        nd = len({x for x in vals if 10 < x < 300})
        n300 = sum(x == 300 for x in vals)
        if m < MARGIN or nd < 6 or n300 < 2 or n300 > 12:
            return None
        return (nd, m)

    B, T = 32, 29
    gum = synth.synth_gumbel(B * mg.min_T(T), tag="bench")
    full = np.load(os.path.join(HERE, "inference_lrw_b32_full.npz"))
    assert np.array_equal(full["gumbel"], gum.numpy()), "bench noise differs from the committed fixture's"
    case("stop_lrw_b32.npz", B, T, synth.synth_video(B, T, tag="bench"), synth.synth_speaker_embedding(B, tag="bench"), gum, accept32)

    # ---- B=2 (config 1): one clip stops mid-sequence, the other never
    def accept2(ln, m):
        a, b = sorted(ln.tolist())
        if m < MARGIN or not (40 <= a <= 260 and b == 300):
            return None
        return (m,)

    g2 = np.load(os.path.join(HERE, "inference_lrw_b2.npz"))
    gum2 = torch.from_numpy(g2["gumbel"])

    def extra2(mel, attn):
        assert np.array_equal(mel.numpy(), g2["mel_post"]) and np.array_equal(attn[:, ::50].numpy(), g2["attn_rows"]), "B=2 run differs from inference_lrw_b2.npz"
        return dict(attn=attn.numpy())

    case("stop_lrw_b2.npz", 2, 29, synth.synth_video(2, 29, tag="video-lrw2"), synth.synth_speaker_embedding(2, tag="spk-lrw2"), gum2, accept2, extra2)


if __name__ == "__main__":
    main()
