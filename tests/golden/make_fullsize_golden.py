#!/usr/bin/env python
"""Golden vectors of THE REFERENCE at the benchmark's full size (BASELINE.json configs[1]: B=32, T=29, S=300).

Same method as make_goldens.py (the reference's VideoExtractor / Decoder imported by file path in the build container, the repo's
deterministic synthetic checkpoint loaded into them, Gumbel noise fed explicitly).  The inputs are bench.py's own batch (synth tags
"bench"), regenerated from integers on every host; stored are the noise and the reference outputs - the full post-net mel of four
clips, the per-frame mean over the 80 mel bins of ALL 32 clips, output lengths, and the attention argmax with its top-2 margin
(`inference_lrw_b32_full.npz`, unchanged since round 1) - and, in files of their own (`*_mel.npz`, `inference_grid_b16_full.npz`,
`inference_avspeech_b32_full.npz`):

  * configs[1]  the FULL (32,80,300) post-net mel of the LRW batch;
  * configs[3]  GRID-shaped: B=16 clips of T in [25,75] frames zero-padded to the batch maximum exactly as the collate pads them
                (datasets/__init__.py:7-46; the model ignores lengths, so the padding is part of the semantics): `inference` (S=300)
                and `forward(tf_ratio=1)` with S = 16000*75/25/256 + 1 = 188 target frames, full tensors;
  * configs[4]  AVSpeech-shaped: B=32 clips of T in [25,50] zero-padded to 50, conditioned on speaker embeddings that come out of the
                SpeakerEncoder route (oracle restatement of audio.py:110-150 on synthetic audio; its torchaudio front-end is parity-
                unpinned, so the embedding is STORED and fed to reference and HIP path alike): `inference` (S=300), full tensors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fullsize_golden.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg          # noqa: E402  (load_reference, GumbelFeed, top2, sub, min_T)

from lip2speech_amd import synth   # noqa: E402
from oracle import l2s_oracle as orc   # noqa: E402

CLIPS = [0, 7, 19, 31]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(16)
    vid, dec = mg.load_reference()
    sd = synth.synth_state_dict()
    enc_ref = vid.VideoExtractor().eval()
    dec_ref = dec.Decoder().eval()
    enc_ref.load_state_dict(mg.sub(sd, "encoder."), strict=True)
    dec_ref.load_state_dict(mg.sub(sd, "decoder."), strict=True)
    B, T, S = 32, 29, 300
    video = synth.synth_video(B, T, tag="bench")
    emb = synth.synth_speaker_embedding(B, tag="bench")
    gum = synth.synth_gumbel(B * mg.min_T(T), tag="bench")
    t0 = time.time()
    with torch.no_grad(), mg.GumbelFeed(dec, gum):
        feat = enc_ref(video)
        face = emb.unsqueeze(1).repeat(1, T, 1)
        mel_post, lengths, attn = dec_ref.inference(torch.cat([feat, face], dim=2), face, return_attention_map=True)
    print(f"reference inference at B={B}, T={T}, S={S}: {time.time() - t0:.1f} s; mel_post {tuple(mel_post.shape)}, attn {tuple(attn.shape)}")
    with torch.no_grad():
        o_post, o_len, o_attn = orc.inference(sd, video, emb, gum, S=S)
    print("oracle-vs-reference  mel_post max|d| = %.3e   attention max|d| = %.3e   lengths equal: %s" %
          ((o_post - mel_post).abs().max().item(), (o_attn - attn).abs().max().item(), bool(torch.equal(o_len, lengths))))
    amax, margin = mg.top2(attn)
    o_amax, _ = mg.top2(o_attn)
    print("attention argmax equal:", bool((amax == o_amax).all()), " min top-2 margin:", margin.min().item())
    assert (o_post - mel_post).abs().max().item() < 1e-3
    np.savez_compressed(
        os.path.join(HERE, "inference_lrw_b32_full.npz"),
        gumbel=gum.numpy(), clips=np.asarray(CLIPS), mel_post_clips=mel_post[CLIPS].numpy(),
        mel_post_frame_mean=mel_post.mean(dim=1).numpy() if mel_post.shape[1] == 80 else mel_post.mean(dim=2).numpy(),
        mel_layout=np.asarray(mel_post.shape), output_lengths=lengths.numpy(),
        attn_argmax=amax.numpy().astype(np.int8), attn_margin=margin.numpy().astype(np.float32))
    print("wrote inference_lrw_b32_full.npz", os.path.getsize(os.path.join(HERE, "inference_lrw_b32_full.npz")) // 1024, "KiB")
    np.savez_compressed(os.path.join(HERE, "inference_lrw_b32_full_mel.npz"), mel_post=mel_post.numpy())
    print("wrote inference_lrw_b32_full_mel.npz", os.path.getsize(os.path.join(HERE, "inference_lrw_b32_full_mel.npz")) // 1024, "KiB")
    variable_t_cases(vid, dec, enc_ref, dec_ref, sd)


def variable_t_cases(vid, dec, enc_ref, dec_ref, sd):
    from lip2speech_amd import statespec
    # ---------------- configs[3]: GRID-shaped, B=16, T in [25,75] padded to 75 ----------------
    B, S = 16, 300
    lens = synth.synth_clip_lengths(B, 25, 75, "grid16")
    T = int(lens.max())
    video = synth.synth_padded_video(B, lens, "grid16")
    emb = synth.synth_speaker_embedding(B, tag="grid16")
    gum = synth.synth_gumbel(B * mg.min_T(T), tag="grid16")
    Sf = 16000 * T // 25 // 256 + 1
    mels = synth.synth_mels(B, Sf, tag="grid16")
    t0 = time.time()
    with torch.no_grad(), mg.GumbelFeed(dec, gum):
        feat = enc_ref(video)
        face = emb.unsqueeze(1).repeat(1, T, 1)
        vis = torch.cat([feat, face], dim=2)
        mel_post, lengths, attn = dec_ref.inference(vis, face, return_attention_map=True)
        fwd = dec_ref(vis, face, mels, torch.as_tensor(lens), torch.full((B,), Sf), 1)
    print(f"reference GRID-shaped B={B}, T={T} (clips {lens.min()}..{lens.max()} frames), inference S={S} + forward S={Sf}: {time.time() - t0:.1f} s")
    with torch.no_grad():
        o_post, o_len, o_attn = orc.inference(sd, video, emb, gum, S=S)
    print("oracle-vs-reference  mel_post max|d| = %.3e  lengths equal: %s" % ((o_post - mel_post).abs().max().item(), bool(torch.equal(o_len, lengths))))
    assert (o_post - mel_post).abs().max().item() < 1e-3
    amax, margin = mg.top2(attn)
    np.savez_compressed(os.path.join(HERE, "inference_grid_b16_full.npz"), gumbel=gum.numpy(), clip_frames=lens, mel_post=mel_post.numpy(),
                        output_lengths=lengths.numpy(), attn_argmax=amax.numpy().astype(np.int8), attn_margin=margin.numpy().astype(np.float32),
                        fwd_S=np.asarray(Sf), fwd_mel=fwd[0].numpy(), fwd_mel_post=fwd[1].numpy(), fwd_stop=fwd[2].numpy())
    print("wrote inference_grid_b16_full.npz", os.path.getsize(os.path.join(HERE, "inference_grid_b16_full.npz")) // 1024, "KiB")

    # ---------------- configs[4]: AVSpeech-shaped, B=32, T in [25,50] padded to 50, embeddings from the SpeakerEncoder route ----------------
    B = 32
    lens = synth.synth_clip_lengths(B, 25, 50, "avs32")
    T = int(lens.max())
    video = synth.synth_padded_video(B, lens, "avs32")
    spk_sd = synth.synth_state_dict(statespec.speaker_encoder_spec("speaker_encoder."), seed=99)
    audio = synth.synth_audio(B, 16000 * T // 25, "avs32")
    with torch.no_grad():
        emb = orc.speaker_encoder_inference(spk_sd, audio)
    gum = synth.synth_gumbel(B * mg.min_T(T), tag="avs32")
    t0 = time.time()
    with torch.no_grad(), mg.GumbelFeed(dec, gum):
        feat = enc_ref(video)
        face = emb.unsqueeze(1).repeat(1, T, 1)
        mel_post, lengths, attn = dec_ref.inference(torch.cat([feat, face], dim=2), face, return_attention_map=True)
    print(f"reference AVSpeech-shaped B={B}, T={T} (clips {lens.min()}..{lens.max()} frames), inference S={S}: {time.time() - t0:.1f} s")
    with torch.no_grad():
        o_post, o_len, o_attn = orc.inference(sd, video, emb, gum, S=S)
    print("oracle-vs-reference  mel_post max|d| = %.3e  lengths equal: %s" % ((o_post - mel_post).abs().max().item(), bool(torch.equal(o_len, lengths))))
    assert (o_post - mel_post).abs().max().item() < 1e-3
    amax, margin = mg.top2(attn)
    np.savez_compressed(os.path.join(HERE, "inference_avspeech_b32_full.npz"), gumbel=gum.numpy(), clip_frames=lens, speaker_embedding=emb.numpy(),
                        mel_post=mel_post.numpy(), output_lengths=lengths.numpy(), attn_argmax=amax.numpy().astype(np.int8),
                        attn_margin=margin.numpy().astype(np.float32))
    print("wrote inference_avspeech_b32_full.npz", os.path.getsize(os.path.join(HERE, "inference_avspeech_b32_full.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
