"""The oracle against the committed reference outputs (tests/golden/*.npz, produced by
tests/golden/make_goldens.py from the imported reference).  CPU only; no /root/reference needed."""
import os

import numpy as np
import pytest
import torch

from lip2speech_amd import statespec, synth
from oracle import l2s_oracle as orc


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def _top2(a):
    srt, idx = torch.sort(a, dim=-1, descending=True)
    return idx[..., 0].to(torch.int32), srt[..., 0] - srt[..., 1]


def test_state_spec_counts():
    enc = statespec.encoder_spec("encoder.")
    dec = statespec.decoder_spec("decoder.")
    assert len(enc) == 337 and len(dec) == 191            # key counts of the reference's two state_dicts
    n_enc = sum(int(np.prod(s)) for _, s, k in enc if k not in statespec.BUFFER_KINDS)
    n_dec = sum(int(np.prod(s)) for _, s, k in dec if k not in statespec.BUFFER_KINDS)
    assert n_enc == 1_151_324 and n_dec == 37_285_512      # SURVEY.md §8(a) parameter totals


def test_synth_is_integer_exact():
    u = synth.uniform01("probe", 5)
    assert u.dtype == np.float32
    # pinned values: the generator must never drift (goldens depend on it)
    np.testing.assert_array_equal(u * 16777216.0, np.round(u * 16777216.0))
    a = synth.synth_video(1, 2, 8, 8, tag="pin")
    assert abs(float(a.std()) - 1.0) < 0.1


def test_oracle_inference_matches_reference_golden(golden_dir, synth_sd):
    g = _load(golden_dir, "inference_lrw_b2.npz")
    B, T = 2, 29
    video = synth.synth_video(B, T, tag="video-lrw2")
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2")
    taps = {}
    with torch.no_grad():
        mel_post, lengths, attn = orc.inference(synth_sd, video, emb, g["gumbel"], S=300, taps=taps)
    assert (taps["feat"] - g["feat"]).abs().max() < 1e-5
    assert (mel_post - g["mel_post"]).abs().max() < 1e-3          # north-star tolerance; observed ~2e-5
    assert torch.equal(lengths, g["output_lengths"])
    amax, _ = _top2(attn)
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure], g["attn_argmax"][sure])
    assert (attn[:, ::50] - g["attn_rows"]).abs().max() < 1e-3


@pytest.mark.parametrize("name,B,T,S,tag", [
    ("forward_lrw_b2_s77.npz", 2, 29, 77, "lrw2"),
    ("forward_grid_b2_t75_s188.npz", 2, 75, 188, "grid2"),
    ("forward_pad_b2_t50_s128.npz", 2, 50, 128, "pad2"),
])
def test_oracle_forward_matches_reference_golden(golden_dir, synth_sd, name, B, T, S, tag):
    g = _load(golden_dir, name)
    video = synth.synth_video(B, T, tag=f"video-{tag}")
    if tag == "pad2":
        video[0, :, 25:] = 0
    emb = synth.synth_speaker_embedding(B, tag=f"spk-{tag}")
    mels = synth.synth_mels(B, S, tag=f"mel-{tag}")
    gum = g["gumbel"] if "gumbel" in g else _load(golden_dir, "inference_lrw_b2.npz")["gumbel"]
    with torch.no_grad():
        out = orc.forward_eval(synth_sd, video, emb, mels, gum)
    assert (out[0] - g["mel"]).abs().max() < 1e-4
    assert (out[1] - g["mel_post"]).abs().max() < 1e-3
    assert (out[2] - g["stop"]).abs().max() < 1e-4
    scale = g["attn_logits"].abs().max()
    assert ((out[4] - g["attn_logits"]).abs().max() / scale) < 1e-5
    assert (out[5] - g["content_dis"]).abs().max() < 1e-6


def test_oracle_teacher_forcing_golden(golden_dir, synth_sd):
    g = _load(golden_dir, "forward_lrw_b2_s77_tf05.npz")
    gum = _load(golden_dir, "inference_lrw_b2.npz")["gumbel"]
    video = synth.synth_video(2, 29, tag="video-lrw2")
    emb = synth.synth_speaker_embedding(2, tag="spk-lrw2")
    mels = synth.synth_mels(2, 77, tag="mel-lrw2")
    with torch.no_grad():
        out = orc.forward_eval(synth_sd, video, emb, mels, gum, teacher_mask=g["teacher_mask"])
    assert g["teacher_mask"].sum() > 10
    assert (out[0] - g["mel"]).abs().max() < 1e-4
    assert (out[1] - g["mel_post"]).abs().max() < 1e-3


def test_oracle_matches_reference_at_benchmark_size(golden_dir, synth_sd):
    """B=32, T=29, S=300 (bench.py's batch, BASELINE.json configs[1]) against tests/golden/make_fullsize_golden.py's reference run."""
    g = _load(golden_dir, "inference_lrw_b32_full.npz")
    video = synth.synth_video(32, 29, tag="bench")
    emb = synth.synth_speaker_embedding(32, tag="bench")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        mel_post, lengths, attn = orc.inference(synth_sd, video, emb, g["gumbel"], S=300)
    clips = [int(c) for c in g["clips"]]
    assert (mel_post[clips] - g["mel_post_clips"]).abs().max() < 1e-3
    assert (mel_post - _load(golden_dir, "inference_lrw_b32_full_mel.npz")["mel_post"]).abs().max() < 1e-3      # every clip, every value
    assert (mel_post.mean(dim=1) - g["mel_post_frame_mean"]).abs().max() < 1e-3
    assert torch.equal(lengths, g["output_lengths"])
    amax, _ = _top2(attn)
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure].to(torch.int64), g["attn_argmax"][sure].to(torch.int64))


@pytest.mark.parametrize("name,B,lo,hi,tag", [("inference_grid_b16_full.npz", 16, 25, 75, "grid16"), ("inference_avspeech_b32_full.npz", 32, 25, 50, "avs32")])
def test_oracle_matches_reference_on_variable_length_batches(golden_dir, synth_sd, name, B, lo, hi, tag):
    """BASELINE.json configs[3] / [4] shapes (clips of different lengths zero-padded by the collate) against the reference run at full size.
    The oracle runs the first 48 decode steps only (CPU minutes): pre-post-net frames do not depend on later steps, and post-net frame t
    sees frames t-10..t+10, so the first 38 post-net frames are complete."""
    g = _load(golden_dir, name)
    lens = synth.synth_clip_lengths(B, lo, hi, tag)
    assert list(lens) == list(g["clip_frames"].numpy())
    video = synth.synth_padded_video(B, lens, tag)
    emb = g["speaker_embedding"] if "speaker_embedding" in g else synth.synth_speaker_embedding(B, tag=tag)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    S = 48
    with torch.no_grad():
        mel_post, _, attn = orc.inference(synth_sd, video, emb, g["gumbel"], S=S)
    assert (mel_post[:, :, :S - 10] - g["mel_post"][:, :, :S - 10]).abs().max() < 1e-3
    amax, _ = _top2(attn)
    sure = g["attn_margin"][:, :S] > 1e-4
    assert torch.equal(amax[sure].to(torch.int64), g["attn_argmax"][:, :S][sure].to(torch.int64))


@pytest.mark.parametrize("name,B,vtag,etag", [("stop_lrw_b2.npz", 2, "video-lrw2", "spk-lrw2"), ("stop_lrw_b32.npz", 32, "bench", "bench")])
def test_oracle_stop_bookkeeping_matches_reference_golden(golden_dir, synth_sd, name, B, vtag, etag):
    """decoder.py:429-435 with first crossings spread over the 300 steps and clips that never stop (tests/golden/make_stop_goldens.py:
    a checkpoint that differs only in the stop layer): the oracle's int64 lengths equal the reference's, its stop logits follow."""
    g = _load(golden_dir, name)
    lens = g["output_lengths"]
    assert lens.dtype == torch.int64 and int((lens == 300).sum()) >= (2 if B > 2 else 1)
    assert B == 2 or len({int(x) for x in lens if 10 < int(x) < 300}) >= 5
    sd = dict(synth_sd)
    sd["decoder.stop_token_layer.linear_layer.weight"] = g["stop_weight"]
    sd["decoder.stop_token_layer.linear_layer.bias"] = g["stop_bias"]
    taps = {}
    with torch.no_grad():
        _, lengths, attn = orc.inference(sd, synth.synth_video(B, 29, tag=vtag), synth.synth_speaker_embedding(B, tag=etag), g["gumbel"], S=300, taps=taps)
    assert lengths.dtype == torch.int64 and torch.equal(lengths, lens)
    assert (taps["stop"].reshape(B, 300) - g["stop_logits"]).abs().max() < 1e-3
    if "attn" in g:
        assert (attn - g["attn"]).abs().max() < 1e-3           # the full (2,300,29) post-softmax attention of the reference
