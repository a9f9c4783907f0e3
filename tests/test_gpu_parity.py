"""Parity of the HIP path (through the C-ABI) against the reference goldens and the CPU oracle.  `-m gpu`.

Tolerances: the north star asks |d| < 1e-3 fp32 on mel frames and identical attention argmax; intermediate
tensors are held to much tighter bounds so a regression is caught where it starts."""
import numpy as np
import pytest
import torch

from lip2speech_amd import native, synth
from oracle import l2s_oracle as orc
import parity_common as pc

pytestmark = pytest.mark.gpu

MEL_TOL = 1e-3


@pytest.fixture(scope="module")
def nm(synth_sd):
    return pc.native_model(synth_sd)


@pytest.fixture(params=["launch", "shipped"])
def nm_both(request, synth_sd):
    """Both arithmetic routes of a small single-batch call: "launch" = the suite's pinned launch-per-phase loop (persist_decode = 0), "shipped" = the
    options the library ships with (persist_decode = 4: persistent decode loop for <= 4 clips of <= 32 frames, persistent BiLSTM for <= 2 clips)."""
    return pc.native_model(synth_sd) if request.param == "launch" else pc.shipped_model(synth_sd)


@pytest.fixture(params=[0, pc.SHIPPED_PERSIST], ids=["launch", "shipped"])
def persist_default(request):
    """The process default of "persist_decode" while the test runs (models the test creates - get_network too - start with it)."""
    with pc.process_default("persist_decode", request.param):
        yield request.param


@pytest.mark.parametrize("M,N,K,act", [(64, 64, 32, 0), (100, 70, 36, 1), (928, 512, 1024, 2), (37, 501, 256, 3),
                                       (130, 58, 58, 1), (9, 80, 2560, 0), (1, 1, 4, 0)])
def test_gemm_operator(M, N, K, act):
    torch.manual_seed(M * 7 + N)
    A = torch.randn(M, K).cuda()
    W = (torch.randn(N, K) / K ** 0.5).cuda()
    sc, sh, aw = (torch.rand(N) + 0.5).cuda(), torch.randn(N).cuda(), (torch.rand(N) + 0.5).cuda()
    C = native.op_gemm(A, W, sc, sh, aw, act)
    ref = (A.double() @ W.double().t()) * sc.double() + sh.double()
    ref = [ref, ref.relu(), ref * torch.sigmoid(ref), torch.sin(ref) * aw.double()][act]
    assert pc.maxdiff(C, ref) < 2e-5


@pytest.mark.parametrize("B,T,Ci,Co,k,st,pad", [(2, 29, 512, 512, 11, 1, 5), (2, 29, 512, 512, 7, 7, 0), (3, 40, 80, 512, 5, 1, 2),
                                                 (2, 75, 512, 80, 5, 1, 2), (2, 29, 512, 512, 3, 3, 0), (1, 7, 512, 512, 7, 7, 0)])
def test_conv1d_operator(B, T, Ci, Co, k, st, pad):
    torch.manual_seed(T + k)
    X = torch.randn(B, T, Ci)
    Wt = torch.randn(Co, Ci, k) / (Ci * k) ** 0.5
    Wp = Wt.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()
    out = native.op_conv1d(X.cuda(), Wp.cuda(), taps=k, stride=st, pad=pad)
    ref = torch.nn.functional.conv1d(X.double().permute(0, 2, 1), Wt.double(), stride=st, padding=pad).permute(0, 2, 1)
    assert pc.maxdiff(out, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,act", [(9600, 512, 2560, 3), (300, 200, 64, 0), (130, 129, 36, 2), (257, 384, 1028, 1)])
def test_gemm_split_bf16_operator(M, N, K, act):
    """gemm_x3.hip: fp32 operands split exactly into three bf16 planes, six bf16 MFMAs per K step.  Gate = the f32-MFMA kernel's own gate;
    and its error against the fp64 product must not exceed 1.5x the f32 kernel's (measured: 0.85-1.0x)."""
    torch.manual_seed(M + N)
    A = torch.randn(M, K).cuda()
    W = (torch.randn(N, K) / K ** 0.5).cuda()
    sc, sh, aw = (torch.rand(N) + 0.5).cuda(), torch.randn(N).cuda(), (torch.rand(N) + 0.5).cuda()
    ref = (A.double() @ W.double().t()) * sc.double() + sh.double()
    ref = [ref, ref.relu(), ref * torch.sigmoid(ref), torch.sin(ref) * aw.double()][act]
    e3 = pc.maxdiff(native.op_gemm(A, W, sc, sh, aw, act, x3=True), ref)
    e32 = pc.maxdiff(native.op_gemm(A, W, sc, sh, aw, act), ref)
    assert e3 < 2e-5 and e3 <= 1.5 * e32 + 1e-7, (e3, e32)


@pytest.mark.parametrize("M,N,K,act", [(9600, 512, 2560, 3), (1000, 256, 1040, 0), (129, 768, 48, 1), (333, 512, 20, 0)])
def test_gemm_split_bf16_tiles_agree(M, N, K, act):
    """the 128x256x16 tile (eight MFMA waves; default where N is a multiple of 256) and the 128x128x32 tile accumulate every output in the same
    order: bit-identical, ragged M and K (odd number of K steps, K tail of 4) included"""
    torch.manual_seed(M + K)
    A = torch.randn(M, K).cuda()
    W = (torch.randn(N, K) / K ** 0.5).cuda()
    sc, sh, aw = (torch.rand(N) + 0.5).cuda(), torch.randn(N).cuda(), (torch.rand(N) + 0.5).cuda()
    wide = native.op_gemm(A, W, sc, sh, aw, act, x3=True)
    narrow = native.op_gemm(A, W, sc, sh, aw, act, x3=True, x3_narrow=True)
    assert torch.equal(wide, narrow)
    assert torch.equal(wide, native.op_gemm(A, W, sc, sh, aw, act, x3=True, x3_dma=True))      # weight planes by LDS-DMA (flags bit 8; wide tile only): same bits
    ref = (A.double() @ W.double().t()) * sc.double() + sh.double()
    ref = [ref, ref.relu(), ref * torch.sigmoid(ref), torch.sin(ref) * aw.double()][act]
    assert pc.maxdiff(wide, ref) < 2e-5


@pytest.mark.parametrize("B,T,Ci,Co,k,st,pad", [(32, 300, 512, 512, 5, 1, 2), (2, 29, 512, 512, 11, 1, 5), (3, 40, 80, 512, 5, 1, 2), (2, 29, 512, 512, 3, 3, 0)])
def test_conv1d_split_bf16_operator(B, T, Ci, Co, k, st, pad):
    torch.manual_seed(T + k)
    X = torch.randn(B, T, Ci)
    Wt = torch.randn(Co, Ci, k) / (Ci * k) ** 0.5
    Wp = Wt.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()
    out = native.op_conv1d(X.cuda(), Wp.cuda(), taps=k, stride=st, pad=pad, x3=True)
    ref = torch.nn.functional.conv1d(X.double().permute(0, 2, 1), Wt.double(), stride=st, padding=pad).permute(0, 2, 1)
    assert pc.maxdiff(out, ref) < 2e-5
    assert torch.equal(out, native.op_conv1d(X.cuda(), Wp.cuda(), taps=k, stride=st, pad=pad, x3=True, x3_narrow=True))      # both tiles: same bits
    assert torch.equal(out, native.op_conv1d(X.cuda(), Wp.cuda(), taps=k, stride=st, pad=pad, x3=True, x3_dma=True))         # and with the weights by LDS-DMA


@pytest.mark.parametrize("M,N,K", [(9600, 512, 2560), (300, 200, 64), (130, 129, 36)])
def test_gemm_bf16_operand_flag(M, N, K):
    """flags bit 1 of l2s_op_gemm_ex (what a model with "infer_bf16" / "train_bf16" runs): operands rounded to nearest even to bf16, fp32
    accumulation - i.e. the fp64 product of the ROUNDED operands up to fp32 summation error; and really not the fp32 product."""
    torch.manual_seed(M + K)
    A = torch.randn(M, K).cuda()
    W = (torch.randn(N, K) / K ** 0.5).cuda()
    out = native.op_gemm(A, W, bf16=True)
    assert pc.maxdiff(out, A.bfloat16().double() @ W.bfloat16().double().t()) < 2e-5
    assert pc.maxdiff(out, A.double() @ W.double().t()) > 1e-4


@pytest.mark.parametrize("hw,T", [(96, 4), (88, 3)])
def test_frontend_kernel_bf16_leg(synth_sd, hw, T):
    """frontend3d_x3_kernel<HW,1>: one bf16 plane.  Equals the oracle's fused front-end fed the bf16-rounded frames and conv weights
    (BatchNorm, PReLU and the pool stay fp32)."""
    b16 = pc.fresh_native_model(synth_sd, infer_bf16=1)
    v = synth.synth_video(1, T, hw, hw, tag=f"fe{hw}")
    sd = dict(synth_sd)
    sd["encoder.frontend3D.0.weight"] = synth_sd["encoder.frontend3D.0.weight"].bfloat16().float()
    out = b16.op_frontend(v.cuda())
    assert pc.maxdiff(out, orc.frontend3d(v.bfloat16().float(), sd).permute(0, 2, 3, 1)) < 2e-5
    assert pc.maxdiff(out, orc.frontend3d(v, synth_sd).permute(0, 2, 3, 1)) > 1e-4


@pytest.mark.parametrize("hw,T", [(96, 4), (88, 3), (96, 1), (96, 7)])
def test_frontend_kernel(nm, synth_sd, hw, T):
    """default: the two-output-frames-per-block split-bf16 kernel (odd T: the last block carries one frame; T = 1: every temporal tap but the
    centre one is padding)"""
    v = synth.synth_video(2, T, hw, hw, tag=f"fe{hw}")
    out = nm.op_frontend(v.cuda())
    ref = orc.frontend3d(v, synth_sd).permute(0, 2, 3, 1)
    assert pc.maxdiff(out, ref) < 2e-5


@pytest.mark.parametrize("x3", [1, 0])
def test_frontend_kernel_other_forms(synth_sd, nm, x3):
    """option frontend_x3 = 1: one output frame per block (32-wide tiles); 0: the f32 MFMA kernel.  Both against the oracle and against the default form."""
    own = pc.fresh_native_model(synth_sd, frontend_x3=x3)
    v = synth.synth_video(2, 5, 96, 96, tag="fe_forms")
    out = own.op_frontend(v.cuda())
    assert pc.maxdiff(out, orc.frontend3d(v, synth_sd).permute(0, 2, 3, 1)) < 2e-5
    d = pc.maxdiff(out, nm.op_frontend(v.cuda()))
    assert 0 < d < 5e-5                                      # another kernel really ran: rounding-level differences, not zero


@pytest.mark.parametrize("hw,T", [(96, 5), (88, 7), (96, 1)])
def test_frontend_interleaved_staging_same_bits(synth_sd, nm, hw, T):
    """option frontend_x3 = 3 (default: the next slab's split + LDS stores interleaved between the MFMA groups, double-buffered input planes) against 2 (a staging
    phase of its own between two barriers): the same operand bits through the same MFMAs - the same output bits, odd T and 88x88 crops included."""
    two = pc.fresh_native_model(synth_sd, frontend_x3=2)
    v = synth.synth_video(2, T, hw, hw, tag=f"fe_pipe{hw}_{T}")
    a, b = nm.op_frontend(v.cuda()), two.op_frontend(v.cuda())
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert pc.maxdiff(a, orc.frontend3d(v, synth_sd).permute(0, 2, 3, 1)) < 2e-5


def test_encoder_matches_reference_golden(nm):
    g, video, _ = pc.lrw2_inputs()
    feat = nm.encoder_fwd(video.cuda())
    assert pc.maxdiff(feat, g["feat"]) < 1e-5
    norms = feat.norm(dim=2)
    assert (norms - 1).abs().max() < 1e-5                       # F.normalize over the 768 channels


@pytest.mark.parametrize("hw,B,T", [(88, 1, 3), (96, 1, 5), (88, 2, 2)])
def test_encoder_other_geometries(nm, synth_sd, hw, B, T):
    """88x88 crops (11x11 at stage 2) and odd frame counts (a half-filled last block of the fused ShuffleNet units)."""
    v = synth.synth_video(B, T, hw, hw, tag=f"enc{hw}_{B}_{T}")
    feat = nm.encoder_fwd(v.cuda())
    ref = orc.encoder_forward(synth_sd, v)
    assert pc.maxdiff(feat, ref) < 2e-5


@pytest.mark.parametrize("hw,B,T", [(96, 2, 29), (88, 1, 3), (96, 1, 5)])
def test_encoder_trunk_f32_units(nm, synth_sd, hw, B, T):
    """option trunk_x3 = 0: the fused ShuffleNet units with f32-MFMA pointwise convs (the default runs them on the bf16 matrix cores through the exact
    three-way split).  Both against the oracle, and against each other at rounding level - another kernel really ran."""
    own = pc.fresh_native_model(synth_sd, trunk_x3=0)
    v = synth.synth_video(B, T, hw, hw, tag=f"trunk{hw}_{B}_{T}")
    feat = own.encoder_fwd(v.cuda())
    assert pc.maxdiff(feat, orc.encoder_forward(synth_sd, v)) < 2e-5
    d = pc.maxdiff(feat, nm.encoder_fwd(v.cuda()))
    assert 0 < d < 2e-6


@pytest.mark.parametrize("hw,B,T", [(96, 2, 29), (88, 1, 3), (96, 1, 5), (96, 71, 29)])
def test_encoder_stage_chain_same_bits(nm, synth_sd, hw, B, T):
    """Round 6: the stride-1 units of a ShuffleNet stage as ONE launch (shuffle_s1xc_kernel: the map stays on chip between the units, the channel
    shuffle is re-cut into the next unit's halves with wave shuffles; diagnostic switch "trunk_chain" = 0 gives one launch per unit).  Per pixel the
    arithmetic is the single-unit kernel's: the features must be the same BITS - also with a half-filled last block (odd frame counts, 88x88
    crops) and with the five-frames-per-block instance of the 3x3 stage (>= 2048 frames in the launch)."""
    per_unit = pc.fresh_native_model(synth_sd, trunk_chain=0)
    v = synth.synth_video(2, T, hw, hw, tag=f"chain{hw}_{T}").repeat((B + 1) // 2, 1, 1, 1, 1)[:B].clone()
    if B > 2:
        v += 0.01 * torch.randn(v.shape, generator=torch.Generator().manual_seed(B))
    a = nm.encoder_fwd(v.cuda())
    b = per_unit.encoder_fwd(v.cuda())
    assert torch.isfinite(a).all() and torch.equal(a, b)
    if B <= 2:
        assert pc.maxdiff(a, orc.encoder_forward(synth_sd, v)) < 2e-5


def test_prologue_matches_oracle(nm_both):
    nm = nm_both
    g, _, emb = pc.lrw2_inputs()
    B, T = 2, 29
    vis = native.build_visual(g["feat"].cuda(), emb.cuda())
    assert pc.maxdiff(vis, orc.build_visual(g["feat"], emb)) == 0.0
    state, dis = nm.decoder_prologue(vis, emb.cuda(), g["gumbel"].cuda())
    sf = lambda f, shape: native.state_field(state, B, T, f, shape)      # noqa: E731
    assert pc.maxdiff(sf(native.ST_ENC, (B, T, 512)), g["oracle_enc"]) < 2e-5
    assert pc.maxdiff(sf(native.ST_K, (B, T, 512)), g["oracle_k"].permute(0, 2, 1)) < 5e-5
    assert pc.maxdiff(sf(native.ST_V, (B, T, 512)), g["oracle_v"]) < 5e-5
    assert pc.maxdiff(sf(native.ST_CKEY, (B, 4, 256)), g["oracle_key"].permute(0, 2, 1)) < 2e-5
    assert pc.maxdiff(sf(native.ST_CVAL, (B, 4, 256)), g["oracle_value"]) < 1e-4
    assert pc.maxdiff(sf(native.ST_ECELL, (B, 512)), g["oracle_encoder_cell"]) < 2e-5
    hfrag = sf(native.ST_H, (2, 16 * 512))
    for layer in range(2):
        assert pc.maxdiff(pc.unfrag(hfrag[layer], B, 512), g["oracle_hidden"][layer]) < 2e-5
    assert pc.maxdiff(dis, pc.golden("forward_lrw_b2_s77.npz")["content_dis"]) < 1e-6


def test_inference_matches_reference_golden(nm_both):
    """The headline parity gate: Lip2Speech.inference, S=300, identical Gumbel noise."""
    nm = nm_both
    g, video, emb = pc.lrw2_inputs()
    mel_post, lengths, attn = nm.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=300, want_attn=True)
    assert pc.maxdiff(mel_post, g["mel_post"]) < MEL_TOL
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure], g["attn_argmax"][sure]), "attention argmax differs from the reference"
    assert pc.maxdiff(attn[:, ::50], g["attn_rows"]) < MEL_TOL


def test_device_normalise_pad_is_bit_identical_to_host_collate():
    """Data boundary on the device (l2s_normalise_pad_frames): the SAMPLE_LRW clips and a ragged synthetic batch of every byte value give
    exactly the fp32 batch the host transforms + collate build (dataset.py:83-86, datasets/__init__.py:7-46)."""
    import os
    from lip2speech_amd.datasets import PackedFrames, train_collate_fn_pad
    from lip2speech_amd.datasets.lrw import load_frames, normalise_mouth
    sample = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sample_lrw")
    clips = [torch.from_numpy(load_frames(os.path.join(sample, f"ABOUT_0000{i}_mouth.npz"))) for i in (1, 2)]
    g = torch.Generator().manual_seed(3)
    clips += [torch.randint(0, 256, (t, 96, 96, 3), dtype=torch.uint8, generator=g) for t in (7, 25, 29, 12)]
    clips.append(torch.arange(256, dtype=torch.uint8).repeat(96 * 96 * 3 * 2 // 256).view(2, 96, 96, 3))      # every byte value
    items = [(normalise_mouth(c.numpy()), torch.zeros(1, 8), torch.zeros(80, 2), torch.zeros(2, 3, 4, 4)) for c in clips]
    want = train_collate_fn_pad(items)[0][0]
    got = PackedFrames(clips).to_device()
    assert got.shape == want.shape == (len(clips), 3, 29, 96, 96)
    assert torch.equal(got.cpu(), want)
    longer = PackedFrames(clips[:2]).to_device(T=31)                      # explicit T: two more zero frames
    assert torch.equal(longer[:, :, :29].cpu(), want[:2]) and float(longer[:, :, 29:].abs().max()) == 0.0
    big = PackedFrames([clips[0]] * 70).to_device()                       # more clips than one launch's table holds
    assert torch.equal(big[69].cpu(), want[0]) and torch.equal(big[64].cpu(), want[0])
    small = PackedFrames([torch.randint(0, 256, (5, 88, 88, 3), dtype=torch.uint8, generator=g)])
    assert torch.equal(small.to_device().cpu(), normalise_mouth(small.data.view(5, 88, 88, 3).numpy()).permute(1, 0, 2, 3).unsqueeze(0))


def test_full_size_lrw_every_clip(nm):
    """BASELINE.json configs[1] in full: all 32 x 80 x 300 post-net mel values of the reference run at the benchmark's own size."""
    g = pc.golden("inference_lrw_b32_full_mel.npz")
    B, T, S = 32, 29, 300
    gum = pc.golden("inference_lrw_b32_full.npz")["gumbel"]
    mel_post, _, _ = nm.inference(synth.synth_video(B, T, tag="bench").cuda(), synth.synth_speaker_embedding(B, tag="bench").cuda(), gum.cuda(), S=S)
    per_clip = (mel_post.cpu().double() - g["mel_post"].double()).abs().amax(dim=(1, 2))
    assert per_clip.max().item() < MEL_TOL, per_clip


@pytest.mark.parametrize("graph", [0, 1])
def test_full_size_grid_shaped_matches_reference_golden(synth_sd, graph):
    """BASELINE.json configs[3]: GRID-shaped batch, B=16, clips of 27..75 frames zero-padded to 75 by the collate, the decode loop
    replayed from a captured hipGraph (graph=1) - against the reference run at that size: `inference` (S=300) and the evaluate path
    `forward(tf_ratio=1)` with S=188 target frames, every element."""
    g = pc.golden("inference_grid_b16_full.npz")
    B, S = 16, 300
    lens = synth.synth_clip_lengths(B, 25, 75, "grid16")
    assert list(lens) == list(g["clip_frames"].numpy()) and int(lens.max()) == 75
    video = synth.synth_padded_video(B, lens, "grid16").cuda()
    emb = synth.synth_speaker_embedding(B, tag="grid16").cuda()
    own = pc.fresh_native_model(synth_sd, use_graph=graph)
    for _ in range(1 + graph):                              # second call = graph replay
        mel_post, lengths, attn = own.inference(video, emb, g["gumbel"].cuda(), S=S, want_attn=True)
    assert pc.maxdiff(mel_post, g["mel_post"]) < MEL_TOL
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure], g["attn_argmax"][sure].to(torch.int32))
    Sf = int(g["fwd_S"])
    assert Sf == 188
    out = own.forward_eval(video, emb, g["gumbel"].cuda(), Sf)
    assert pc.maxdiff(out[0], g["fwd_mel"]) < MEL_TOL and pc.maxdiff(out[1], g["fwd_mel_post"]) < MEL_TOL
    assert pc.maxdiff(out[2].reshape(B, Sf), g["fwd_stop"].reshape(B, Sf)) < 1e-3


def test_full_size_avspeech_shaped_matches_reference_golden(synth_sd, nm):
    """BASELINE.json configs[4] (fp32 leg): AVSpeech-shaped batch, B=32, clips of 26..50 frames zero-padded to 50, conditioned on
    SpeakerEncoder embeddings.  The embedding route runs on the HIP path and is checked against the stored oracle embedding (its mel
    front-end restates torchaudio: parity unpinned, so the mel gate below uses the STORED embedding the reference was fed)."""
    from lip2speech_amd import statespec
    from lip2speech_amd.model.modules import SpeakerEncoder
    g = pc.golden("inference_avspeech_b32_full.npz")
    B, S = 32, 300
    lens = synth.synth_clip_lengths(B, 25, 50, "avs32")
    assert list(lens) == list(g["clip_frames"].numpy()) and int(lens.max()) == 50
    video = synth.synth_padded_video(B, lens, "avs32").cuda()
    spk_sd = synth.synth_state_dict(statespec.speaker_encoder_spec("speaker_encoder."), seed=99)
    spk = SpeakerEncoder(state_dict={k[len("speaker_encoder."):]: v for k, v in spk_sd.items()}).cuda()
    emb_hip = spk.inference(synth.synth_audio(B, 16000 * 50 // 25, "avs32").cuda())
    assert pc.maxdiff(emb_hip, g["speaker_embedding"]) < 2e-4
    # two half-batches in flight would change nothing either: the batch goes through the grouped entry point as 2 x 16 clips
    emb = g["speaker_embedding"].cuda()
    mel_post, lengths, attn = nm.inference(video, emb, g["gumbel"].cuda(), S=S, want_attn=True)
    assert pc.maxdiff(mel_post, g["mel_post"]) < MEL_TOL
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure], g["attn_argmax"][sure].to(torch.int32))
    m = native.min_T(50)
    halves = nm.inference_multi([(video[:16], emb[:16], g["gumbel"][:16 * m].cuda()), (video[16:], emb[16:], g["gumbel"][16 * m:].cuda())], S=S)
    assert torch.equal(torch.cat([halves[0][0], halves[1][0]]), mel_post)


# the bf16 leg (BASELINE.json configs[4] "... bf16"): the 1e-3 fp32 gate does not apply to it (judged on throughput and a
# stated band).  Bands: bf16 operands carry 2^-9 relative rounding; measured on the LRW batch: encoder features 1.5e-3 rms, mel frames
# 5e-3 mean / 4e-2 max absolute on frames of mean magnitude 1.07, flat along the 300 steps (tools/check_bf16_leg.py).  The gates are TWICE
# the measured deviation, so a regression that doubles the bf16 error fails
BF16_MEAN, BF16_MAX = 1e-2, 8e-2


def test_bf16_leg_tracks_the_reference_b2(synth_sd, nm):
    """Option "infer_bf16" on a model of its own: front-end conv, GEMMs and Conv1d stacks with bf16 operands, fp32 loop - against the
    reference's B=2 golden inside the stated band, really different from the fp32 path, and leaving other models alone."""
    g, video, emb = pc.lrw2_inputs()
    args = (video.cuda(), emb.cuda(), g["gumbel"].cuda())
    base = nm.inference(*args, S=300)[0].clone()
    b16 = pc.fresh_native_model(synth_sd, infer_bf16=1)
    feat32, feat16 = nm.encoder_fwd(video.cuda()), b16.encoder_fwd(video.cuda())
    assert 1e-5 < pc.maxdiff(feat16, feat32) < 1e-2                 # unit-norm rows: bf16 front-end + conv_last really ran, inside 2^-7
    mel, lengths = b16.inference(*args, S=300)[:2]
    d = (mel.cpu() - g["mel_post"]).abs()
    assert d.mean().item() < BF16_MEAN and d.max().item() < BF16_MAX
    assert d.max().item() > 1e-3                                   # not the fp32 path
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    assert torch.equal(nm.inference(*args, S=300)[0], base)        # the fp32 model is untouched
    # grouped: the bf16 leg too is one launch chain over G batches with per-batch results independent of the grouping
    grp = b16.inference_multi([args, args], S=300)
    assert torch.equal(grp[0][0], mel) and torch.equal(grp[1][0], mel)


def test_bf16_leg_avspeech_shaped_full_size(synth_sd):
    """BASELINE.json configs[4], bf16 leg at full size: B=32, clips of 26..50 frames zero-padded, stored speaker embeddings - mel inside the
    stated band of the reference's fp32 golden, the same output lengths, attention argmax equal on >= 99.5 % of the positions whose
    top-2 margin exceeds 1e-3."""
    g = pc.golden("inference_avspeech_b32_full.npz")
    B, S = 32, 300
    lens = synth.synth_clip_lengths(B, 25, 50, "avs32")
    video = synth.synth_padded_video(B, lens, "avs32").cuda()
    b16 = pc.fresh_native_model(synth_sd, infer_bf16=1)
    mel_post, lengths, attn = b16.inference(video, g["speaker_embedding"].cuda(), g["gumbel"].cuda(), S=S, want_attn=True)
    d = (mel_post.cpu() - g["mel_post"]).abs()
    assert d.mean().item() < BF16_MEAN and d.max().item() < BF16_MAX
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-3
    agree = (amax[sure] == g["attn_argmax"][sure].to(torch.int32)).float().mean().item()
    print(f"bf16 leg, AVSpeech-shaped B=32: mel mean |d| {d.mean().item():.2e} max {d.max().item():.2e}; attention argmax agreement {agree:.4f} on "
          f"{int(sure.sum())} positions with margin > 1e-3")
    assert agree >= 0.995


@pytest.mark.parametrize("opts,exact", [
    ({"use_graph": 1}, True),              # BASELINE.json configs[3]: the decode loop replayed from a captured hipGraph - same kernels, same order
    ({"overlap_postnet": 1}, False),       # windowed post-net on a second stream: every output frame sees the same taps (at <= 640 rows per batch the default
                                           # post-net adds its five taps as separate K slices, the windows do not: another order of the same sums)
    ({"skinny_static": 1}, True),          # compile-time K-segment layouts: same chunk -> wave assignment and summation order
    ({"skinny_sized": 0}, True), ({"skinny_split": 1}, True), ({"skinny_split": 3}, True), ({"skinny_split8": 2}, True),
    ({"fold_step_weights": 0}, False),     # literal 6-phase step: different (unmerged) weights, same mathematics
    ({"fuse_trunk": 0}, False), ({"fuse_s2": 0}, False),      # unfused ShuffleNet units: GEMM kernel instead of in-LDS MFMA chain
    ({"gemm_x3": 0}, False),               # f32-MFMA GEMMs everywhere (at B=2 every GEMM is below the split-bf16 threshold anyway)
    ({"use_graph": 1, "fold_step_weights": 0}, False),
    ({"hoist_vproj": 1}, False), ({"hoist_vproj": 0}, False),      # attention_proj on the values (K = 1280 layer 0) / on a @ v through W_ih W_ap (K = 1536)
    ({"skinny_flat": 0}, True),            # uniform first-phase grid
    ({"frontend_x3": 2}, True), ({"frontend_x3": 1}, False), ({"frontend_x3": 0}, False),      # front-end conv: staging as a phase of its own (same bits) / one output frame per block / the f32 MFMA kernel
    ({"lstm_x3": 0}, False), ({"lstm_x3": 1}, True),               # LSTM launches on the f32 matrix pipe / the four-wave split-bf16 form
    ({"attn_lds": 0}, True), ({"attn_lds": 2}, True),              # attention blocks: one-column value loads / values through LDS (the default picks by rows)
    ({"skinny_rc_jb": 28}, False),         # the straight-line blocks on eight waves (they do not sum u in the loader: layer 0 on K = 1280, other bits)
])
def test_runtime_options_keep_parity(synth_sd, nm, opts, exact):
    """Every run-time option of include/l2s.h against the reference golden (B=2, S=300, same Gumbel noise), on a model of its own
    (options are per model); where DESIGN.md claims bit-identity with the default path it is asserted."""
    g, video, emb = pc.lrw2_inputs()
    own = pc.fresh_native_model(synth_sd, **opts)
    for _ in range(2):                                     # the second call replays the cached graph / reuses the side stream
        mel_post, lengths, attn = own.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=300, want_attn=True)
    assert pc.maxdiff(mel_post, g["mel_post"]) < MEL_TOL
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure], g["attn_argmax"][sure])
    if exact:
        ref = nm.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=300, want_attn=True)
        assert torch.equal(mel_post, ref[0]) and torch.equal(attn, ref[2])
    # the same options through the staged entry points (l2s_decode_steps is where use_graph lives for evaluate-style callers)
    out = own.forward_eval(video.cuda(), emb.cuda(), g["gumbel"].cuda(), 77)
    gf = pc.golden("forward_lrw_b2_s77.npz")
    assert pc.maxdiff(out[1], gf["mel_post"]) < MEL_TOL


@pytest.mark.parametrize("rows", [256, 224, 192, 512])
def test_half_cu_block_forms_same_bits(synth_sd, nm, rows):
    """Round 5: at >= 192 rows per launch (and several chains in flight) the step's LSTM launches run on four-wave blocks of half a compute unit (a wave plays its two K slices one after
    the other on one accumulator set, option "lstm_x3" = 3), the first phase on four-wave 2x1 / 2x2 blocks ("flat_half") and the attention always in its
    74-register form ("attn_lds" = 2), so that launch chains overlap on the CUs.  Every partial sum is the one the round-4 block forms compute: the decode
    loop's mel frames, stop logits and attention logits are the same bits (and rows 0-1, the golden clips, stay inside the reference's gate)."""
    T, S = 29, 12
    native.set_thread_chains(3)            # the half-CU forms are what a caller with several chains in flight gets (InflightPool's worker threads)
    try:
        _half_cu_forms_same_bits(synth_sd, nm, rows, T, S)
    finally:
        native.set_thread_chains(1)


def _half_cu_forms_same_bits(synth_sd, nm, rows, T, S):
    g, video2, emb2 = pc.lrw2_inputs()
    reps = rows // 2
    video = video2.repeat(reps, 1, 1, 1, 1).clone(); emb = emb2.repeat(reps, 1).clone(); gum = g["gumbel"].view(2, 4, -1).repeat(reps, 1, 1).reshape(rows * 4, -1).clone()
    video[2:] += 0.01 * torch.randn(video[2:].shape, generator=torch.Generator().manual_seed(rows))      # other clips than the golden pair
    feat = nm.encoder_fwd(video.cuda())
    vis = native.build_visual(feat, emb.cuda())
    outs = []
    for opts in ({}, {"lstm_x3": 2, "flat_half": 0, "attn_lds": 1}, {"lstm_x3": 1, "flat_half": 0}, {"lstm_x3": 3, "flat_half": 2, "attn_lds": 2, "half_min_mts": 8}):
        own = pc.fresh_native_model(synth_sd, **opts)
        state, _ = own.decoder_prologue(vis, emb.cuda(), gum.cuda())
        mel, stop, attn = own.decode_steps(state, rows, T, S, attn_logits=True)
        outs.append((mel.clone(), stop.clone(), attn.clone()))
        del own
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
    post, cf = nm.postnet(outs[0][0], want_cf=True)
    assert torch.isfinite(post).all()
    ref = nm.inference(video2.cuda(), emb2.cuda(), g["gumbel"].cuda(), S=S)
    assert pc.maxdiff(post[:2], ref[0]) < 1e-4


@pytest.mark.parametrize("B,S", [(32, 300), (256, 77), (40, 123)])
def test_postnet_weight_planes_by_dma_same_bits(synth_sd, nm, B, S):
    """The post-net's Conv1d weights as pre-split bf16 planes fetched by LDS-DMA (option "gemm_x3_dma", default) against the staging waves' own load +
    split + store of the fp32 weights: the same operand bits, so the same output bits - at sizes that run the 128x256x16 split-bf16 tile (K = 400 with
    an odd number of K steps and K = 2560), and after a device-side weight refresh (the planes are re-derived with the other derived weights)."""
    torch.manual_seed(3)
    mel = torch.randn(B, S, 80, device="cuda")
    off = pc.fresh_native_model(synth_sd, gemm_x3_dma=0)   # a diagnostic switch: this model runs in libl2s_diag.so (same sources as the product library)
    a, _ = nm.postnet(mel)
    b, _ = off.postnet(mel)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    off.set_option("gemm_x3_dma", 1)                      # the switch is read at launch time
    c, _ = off.postnet(mel)
    assert torch.equal(a, c)
    if B == 32:
        # bound weights changed in place, then the device-side refresh: both forms must follow the new weights (stale planes would keep the old output)
        outs = []
        for v_ in (1, 0):
            m_ = native.NativeModel(native.diag())
            m_.set_option("refresh_map", 1); m_.set_option("gemm_x3_dma", v_)
            m_.load({k: v.cuda() for k, v in synth_sd.items()}, list(synth_sd.keys()))
            params = {k: v.cuda().clone() for k, v in synth_sd.items() if v.is_floating_point()}
            m_.train_bind(params, {})
            for k in params:
                if k.startswith("decoder.postnet.convolutions.") and k.endswith("conv.weight"):
                    params[k].mul_(1.25)
            m_.train_refresh_weights()
            outs.append(m_.postnet(mel)[0])
        assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], a)


def test_options_are_per_model(synth_sd, nm):
    """`l2s_set_option` only changes the defaults of models created later; a model's own switches do not leak into another model."""
    a = pc.fresh_native_model(synth_sd, fold_step_weights=0)
    g, video, emb = pc.lrw2_inputs()
    args = (video.cuda(), emb.cuda(), g["gumbel"].cuda())
    base = nm.inference(*args, S=40)[0].clone()
    lit = a.inference(*args, S=40)[0]
    assert pc.maxdiff(lit, base) > 0 and pc.maxdiff(lit, base) < 1e-3          # the literal step really ran on `a` ...
    assert torch.equal(nm.inference(*args, S=40)[0], base)                     # ... and the shared default model is untouched
    native.set_option("fold_step_weights", 0)
    try:
        b = pc.fresh_native_model(synth_sd)                                    # inherits the new default
        assert torch.equal(b.inference(*args, S=40)[0], lit)
        assert torch.equal(nm.inference(*args, S=40)[0], base)                 # existing models keep their own copy
    finally:
        native.set_option("fold_step_weights", 1)
    with pytest.raises(RuntimeError):
        a.set_option("no_such_option", 1)
    with pytest.raises(RuntimeError):
        a.set_option("skinny_rc", 11)          # a block-form A/B switch: the diagnostic build's (include/l2s_diag.h), unknown to the product library


def test_full_size_matches_reference_golden(nm):
    """BASELINE.json's own configuration (B=32, T=29, S=300 - bench.py's batch) against the reference run at that size
    (tests/golden/make_fullsize_golden.py): four clips element by element, all 32 through their per-frame means, lengths, argmax."""
    g = pc.golden("inference_lrw_b32_full.npz")
    B, T, S = 32, 29, 300
    video = synth.synth_video(B, T, tag="bench")
    emb = synth.synth_speaker_embedding(B, tag="bench")
    mel_post, lengths, attn = nm.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=S, want_attn=True)
    assert tuple(mel_post.shape) == tuple(int(x) for x in g["mel_layout"])
    clips = [int(c) for c in g["clips"]]
    assert pc.maxdiff(mel_post[clips], g["mel_post_clips"]) < MEL_TOL
    assert pc.maxdiff(mel_post.mean(dim=1), g["mel_post_frame_mean"]) < MEL_TOL
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    assert sure.float().mean() > 0.99
    assert torch.equal(amax[sure].to(torch.int64), g["attn_argmax"][sure].to(torch.int64)), "attention argmax differs from the reference"


@pytest.mark.parametrize("B,T,HW,S", [(1, 7, 96, 3), (3, 75, 88, 5), (17, 8, 96, 4), (33, 29, 96, 6), (2, 13, 88, 300), (11, 16, 96, 4),
                                      (3, 59, 88, 24), (19, 14, 88, 38)])
def test_shapes_against_oracle(nm_both, synth_sd, B, T, HW, S):
    """Shapes off the benchmark's grid (tools/fuzz_parity.py draws more): tile remainders of every kernel, the shortest and the longest
    clips, both crop sizes, batches that are not multiples of the 16-row tile."""
    tag = f"fz{B}_{T}_{HW}_{S}"
    v = synth.synth_video(B, T, HW, HW, tag=tag)
    e = synth.synth_speaker_embedding(B, tag=tag)
    g = synth.synth_gumbel(B * native.min_T(T), tag=tag)
    nm = nm_both
    mel, ln, at = nm.inference(v.cuda(), e.cuda(), g.cuda(), S=S, want_attn=True)
    with torch.no_grad():
        omel, oln, oat = orc.inference(synth_sd, v, e, g, S=S)
    assert pc.maxdiff(mel, omel) < MEL_TOL
    assert torch.equal(ln.cpu(), oln)
    amax, _ = pc.top2(at.cpu())
    oamax, margin = pc.top2(oat)
    sure = margin > 1e-4
    assert torch.equal(amax[sure], oamax[sure])


def test_model_api_inference(synth_sd, persist_default):
    """Through the boundary the reference's callers use: get_network('test').inference(...) (demo.py:82-86)."""
    from model.model import get_network
    g, video, emb = pc.lrw2_inputs()
    net = get_network("test")
    net.load_state_dict(synth_sd, strict=True)
    net = net.cuda()
    mel, lengths, attn = net.inference(video.cuda(), None, speaker_embedding=emb.cuda(), return_attention_map=True,
                                       gumbel_noise=g["gumbel"].cuda())
    assert mel.shape == (2, 80, 300) and lengths.dtype == torch.int64 and attn.shape == (2, 300, 29)
    assert pc.maxdiff(mel, g["mel_post"]) < MEL_TOL
    # the staged route (net.encoder / net.decoder used separately, as train.py / demo.py may) agrees with the fused call
    feat = net.encoder(video.cuda())
    vis = native.build_visual(feat, emb.cuda())
    mel2, len2 = net.decoder.inference(vis, emb.cuda().unsqueeze(1).expand(-1, 29, -1), gumbel_noise=g["gumbel"].cuda())
    assert pc.maxdiff(mel2, mel) == 0.0 and torch.equal(len2, lengths)
    # without supplied noise the call draws its own (eval-mode stochasticity of the reference, decoder.py:257)
    mel3, _ = net.inference(video.cuda(), None, speaker_embedding=emb.cuda())
    assert torch.isfinite(mel3).all() and pc.maxdiff(mel3, mel) > 0
    # the bf16 leg is a switch on the model's own native handle: on -> inside the stated band, off -> the fp32 result again, bit for bit
    net.native_model().set_option("infer_bf16", 1)
    mel4, _ = net.inference(video.cuda(), None, speaker_embedding=emb.cuda(), gumbel_noise=g["gumbel"].cuda())
    assert 1e-3 < pc.maxdiff(mel4, g["mel_post"]) < BF16_MAX
    net.native_model().set_option("infer_bf16", 0)
    assert torch.equal(net.inference(video.cuda(), None, speaker_embedding=emb.cuda(), gumbel_noise=g["gumbel"].cuda())[0], mel)


@pytest.mark.parametrize("name,B,T,S,tag", [
    ("forward_lrw_b2_s77.npz", 2, 29, 77, "lrw2"),
    ("forward_grid_b2_t75_s188.npz", 2, 75, 188, "grid2"),      # variable-T: min_T = 10
    ("forward_pad_b2_t50_s128.npz", 2, 50, 128, "pad2"),        # zero-padded clip: lengths are ignored
])
def test_forward_eval_matches_reference_golden(synth_sd, persist_default, name, B, T, S, tag):
    """evaluate.py:38 semantics: net(...,tf_ratio=1) in eval mode -> list of 7."""
    from model.model import get_network
    g = pc.golden(name)
    video = synth.synth_video(B, T, tag=f"video-{tag}")
    if tag == "pad2":
        video[0, :, 25:] = 0
    emb = synth.synth_speaker_embedding(B, tag=f"spk-{tag}")
    mels = synth.synth_mels(B, S, tag=f"mel-{tag}")
    gum = g["gumbel"] if "gumbel" in g else pc.golden("inference_lrw_b2.npz")["gumbel"]
    net = get_network("test")
    net.load_state_dict(synth_sd, strict=True)
    net = net.cuda()
    lens = torch.full((B,), T)
    out = net(video.cuda(), None, None, mels.cuda(), lens, None, torch.full((B,), S), 1,
              speaker_embedding=emb.cuda(), gumbel_noise=gum.cuda())
    assert len(out) == 7 and out[6] is lens
    assert pc.maxdiff(out[0], g["mel"]) < 1e-4
    assert pc.maxdiff(out[1], g["mel_post"]) < MEL_TOL
    assert out[2].shape == (B, S, 1) and pc.maxdiff(out[2], g["stop"]) < 1e-4
    assert pc.maxdiff(out[3], emb) == 0.0
    scale = g["attn_logits"].abs().max().item()
    assert pc.maxdiff(out[4], g["attn_logits"]) / scale < 1e-5      # pre-softmax logits, |values| in the thousands
    assert pc.maxdiff(out[5], g["content_dis"]) < 1e-6


def test_teacher_forced_steps(nm_both):
    """Scheduled sampling made explicit: the reference's own torch.rand draws, replayed as a step mask."""
    nm = nm_both
    g = pc.golden("forward_lrw_b2_s77_tf05.npz")
    base, video, emb = pc.lrw2_inputs()
    B, T, S = 2, 29, 77
    mels = synth.synth_mels(B, S, tag="mel-lrw2")
    vis = native.build_visual(base["feat"].cuda(), emb.cuda())
    state, _ = nm.decoder_prologue(vis, emb.cuda(), base["gumbel"].cuda())
    sd = synth.synth_state_dict()
    teacher = torch.cat([sd["decoder.BOS"].expand(B, 1, -1), mels.permute(0, 2, 1)[:, :S - 1]], dim=1).contiguous()
    mel, stop, _ = nm.decode_steps(state, B, T, S, teacher=teacher.cuda(), teacher_mask=g["teacher_mask"].numpy())
    post, cf = nm.postnet(mel, want_cf=True)
    assert pc.maxdiff(cf, g["mel"]) < 1e-4
    assert pc.maxdiff(post, g["mel_post"]) < MEL_TOL


def test_full_batch_rows_are_independent(nm):
    """BASELINE size (B=32, T=29, S=300): clips are independent, so rows 0-1 of a full batch whose first two clips
    are the golden clips must reproduce the reference golden, whatever the other 30 clips are."""
    g, video2, emb2 = pc.lrw2_inputs()
    B = 32
    video = synth.synth_video(B, 29, tag="full").clone()
    emb = synth.synth_speaker_embedding(B, tag="full").clone()
    gum = synth.synth_gumbel(B * 4, tag="full").clone()
    video[:2], emb[:2], gum[:8] = video2, emb2, g["gumbel"]
    mel_post, lengths, attn = nm.inference(video.cuda(), emb.cuda(), gum.cuda(), S=300, want_attn=True)
    assert torch.isfinite(mel_post).all()
    assert pc.maxdiff(mel_post[:2], g["mel_post"]) < MEL_TOL
    assert torch.equal(lengths[:2].cpu(), g["output_lengths"])
    # permutation equivariance over the batch (a checksum-of-rows property at full size)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    gperm = gum.view(B, 4, -1)[perm].reshape(B * 4, -1)
    mel_p, len_p, _ = nm.inference(video[perm].cuda(), emb[perm].cuda(), gperm.cuda(), S=300)
    assert pc.maxdiff(mel_p, mel_post[perm.cuda()]) < 1e-4
    assert torch.equal(len_p, lengths[perm.cuda()])


def test_edge_shapes(nm_both, synth_sd):
    """B=1, the shortest clip the reference accepts (T=7: the stride-7 content branch), S=1, and B not a multiple of 16."""
    nm = nm_both
    for B, T, S in [(1, 7, 1), (3, 12, 5), (17, 9, 3)]:
        video = synth.synth_video(B, T, tag=f"edge{B}")
        emb = synth.synth_speaker_embedding(B, tag=f"edge{B}")
        gum = synth.synth_gumbel(B * native.min_T(T), tag=f"edge{B}")
        mels = synth.synth_mels(B, S, tag=f"edge{B}")
        feat = nm.encoder_fwd(video.cuda())
        state, _ = nm.decoder_prologue(native.build_visual(feat, emb.cuda()), emb.cuda(), gum.cuda())
        mel, stop, attn = nm.decode_steps(state, B, T, S, attn_logits=True)
        post, cf = nm.postnet(mel, want_cf=True)
        with torch.no_grad():
            ref = orc.forward_eval(synth_sd, video, emb, mels, gum)
        assert pc.maxdiff(cf, ref[0]) < 1e-4 and pc.maxdiff(post, ref[1]) < MEL_TOL
        assert pc.maxdiff(stop, ref[2][:, :, 0]) < 1e-4
    with pytest.raises(RuntimeError):                      # T < 7 is rejected like the reference (conv kernel > input)
        nm.decoder_prologue(torch.zeros(1, 6, 1024).cuda(), torch.zeros(1, 256).cuda(), torch.zeros(1, 501).cuda())


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_sample_lrw_clips_plumbing(nm, synth_sd, tmp_path):
    """BASELINE.json configs[0]: all 10 SAMPLE_LRW clips in batches of 2, the way demo.py / evaluate.py feed them - `LRW` loader
    (bz2 pickle of JPEGs -> PIL), `DataLoader`, collate - with the data boundary on the device (uint8 clips -> l2s_normalise_pad_frames) and
    the five batches advanced as ONE launch chain; every batch against the oracle on the host-collated frames (plumbing with real data:
    synthetic weights, supplied embedding; cv2 and PIL may decode a JPEG 1 LSB apart, so this is not a parity gate against the reference)."""
    import os
    import shutil
    from torch.utils.data import DataLoader
    from lip2speech_amd.datasets import LRW, device_collate_fn_pad, train_collate_fn_pad
    root = os.path.join(pc.GOLDEN, "sample_lrw")
    d = tmp_path / "LRW_Faces" / "ABOUT" / "test"
    a = tmp_path / "lipread_audio" / "ABOUT" / "test"
    d.mkdir(parents=True); a.mkdir(parents=True)
    for i in range(1, 11):
        shutil.copy(os.path.join(root, f"ABOUT_{i:05d}_mouth.npz"), d)
        shutil.copy(os.path.join(root, f"ABOUT_{i:05d}.npz"), a)
    raw, ref = LRW(str(tmp_path), mode="test", raw_frames=True), LRW(str(tmp_path), mode="test")
    assert len(raw) == 10
    dev_batches = list(DataLoader(raw, batch_size=2, shuffle=False, collate_fn=device_collate_fn_pad))
    host_batches = list(DataLoader(ref, batch_size=2, shuffle=False, collate_fn=train_collate_fn_pad))
    assert len(dev_batches) == 5
    group = []
    for i, ((packed, vlen), _, (mels, mlen, gate), _) in enumerate(dev_batches):
        video = packed.to_device()
        assert video.shape == (2, 3, 29, 96, 96) and mels.shape == (2, 80, 77) and vlen.tolist() == [29, 29]
        assert torch.equal(video.cpu(), host_batches[i][0][0])                  # device collate == host collate, bit for bit
        group.append((video, synth.synth_speaker_embedding(2, tag=f"sample{i}").cuda(), synth.synth_gumbel(2 * 4, tag=f"sample{i}").cuda()))
    outs = nm.inference_multi(group, S=300)
    for i, (mel_post, lengths, _) in enumerate(outs):
        with torch.no_grad():
            ref_post, ref_len, _ = orc.inference(synth_sd, host_batches[i][0][0], group[i][1].cpu(), group[i][2].cpu(), S=300)
        assert pc.maxdiff(mel_post, ref_post) < MEL_TOL
        assert torch.equal(lengths.cpu(), ref_len)


def test_caller_loops_voice_route(synth_sd):
    """demo.py / evaluate.py model-facing loops on real SAMPLE_LRW clips with the voice tower supplying the embedding."""
    import os
    from torch.utils.data import DataLoader
    from lip2speech_amd import callers, statespec
    from lip2speech_amd.datasets import test_collate_fn_pad, train_collate_fn_pad
    from lip2speech_amd.datasets.lrw import load_frames, normalise_mouth
    from lip2speech_amd.datasets.spectrograms import MelSpectrogram
    from model.model import get_network
    from model.modules import SpeakerEncoder
    root = os.path.join(pc.GOLDEN, "sample_lrw")
    mel_t = MelSpectrogram()
    items = []
    for i in (1, 2):
        mouth = normalise_mouth(load_frames(os.path.join(root, f"ABOUT_0000{i}_mouth.npz")))
        speech = torch.from_numpy(np.load(os.path.join(root, f"ABOUT_0000{i}.npz"))["data"][None])
        items.append((mouth, speech, mel_t(speech).squeeze(0), torch.zeros(2, 3, 160, 160)))
    net = get_network("test")
    net.load_state_dict(synth_sd, strict=True)
    net = net.cuda()
    spk_sd = synth.synth_state_dict(statespec.speaker_encoder_spec("speaker_encoder."), seed=99)
    spk = SpeakerEncoder(state_dict={k[len("speaker_encoder."):]: v for k, v in spk_sd.items()}).cuda()
    # demo: batch_size 1, file paths appended by the test collate
    one = test_collate_fn_pad([items[0] + (("face.npz", "audio.npz"),)])
    two = test_collate_fn_pad([items[1] + (("face.npz", "audio.npz"),)])
    mel, lengths, attn = callers.demo_clip(net, one, speaker_encoder=spk)
    assert net.native_model().calls["l2s_inference"] == 1      # one clip: the direct call (the library's latency form where the device allows it)
    assert mel.shape[0] == 1 and mel.shape[1] == 80 and mel.shape[2] == int(lengths[0]) and attn.shape[2] == 29
    # the loader-driven form runs on the GROUPED path (three clips = one l2s_inference_multi chain) and returns the clips in order
    nm = net.native_model()
    nm.calls.clear()
    clips = list(callers.demo_clips(net, [one, two, one], speaker_encoder=spk, group=8, n_inflight=2))
    assert nm.calls["l2s_inference_multi"] == 1 and nm.calls["l2s_inference"] == 0
    assert len(clips) == 3 and all(c[0].shape[2] == int(c[1][0]) for c in clips)
    assert clips[0][0].shape == clips[2][0].shape and clips[0][2].shape[2] == 29
    # evaluate: B=2 batches, tf_ratio=1 -> S = 77 target frames; three loader batches = ONE l2s_forward_eval_multi chain
    batch = train_collate_fn_pad(items)
    nm.calls.clear()
    outs = callers.evaluate_mels(net, [batch, batch, batch], speaker_encoder=spk)
    assert nm.calls["l2s_forward_eval_multi"] == 1 and nm.calls["l2s_forward_eval"] == 0
    assert len(outs) == 3 and outs[0].shape == (2, 80, 77) and torch.isfinite(outs[0]).all()
    # the Gumbel noise is drawn per batch (decoder.py:257), so identical batches differ in their content path only: same embedding route
    # same embedding computed by the oracle -> same first mel frames through the oracle path
    with torch.no_grad():
        emb_ref = orc.speaker_encoder_inference(spk_sd, torch.cat([it[1] for it in items], dim=0))
    assert pc.maxdiff(spk.inference(torch.cat([it[1] for it in items], dim=0).cuda()), emb_ref) < 2e-4


@pytest.mark.gpu
def test_inflight_pool_is_bit_identical_to_sequential(synth_sd):
    """Chains in flight on several streams and G batches per chain (parallel.InflightPool: ONE shared weight blob, `l2s_inference_multi`)
    give exactly the one-at-a-time results, in order - including a ragged last group and a shape change inside the list."""
    import parity_common as pc
    from lip2speech_amd import synth
    from lip2speech_amd.parallel import InflightPool
    B, T, S = 4, 29, 40
    batches = [(synth.synth_video(B, T, tag=f"pool{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"pool{i}").cuda(),
                synth.synth_gumbel(B * 4, tag=f"pool{i}").cuda()) for i in range(7)]
    batches.append((synth.synth_video(B, 31, tag="pool7").cuda(), synth.synth_speaker_embedding(B, tag="pool7").cuda(),
                    synth.synth_gumbel(B * 4, tag="pool7").cuda()))                 # another T: closes the running group
    nm = pc.native_model(synth_sd)
    want = [nm.inference(*b, S=S, want_attn=True) for b in batches]
    want = [tuple(t.clone() for t in w) for w in want]
    for n_inflight, group in ((3, 1), (2, 3), (1, 8)):
        pool = InflightPool({k: v.cuda() for k, v in synth_sd.items()}, n_inflight=n_inflight, group=group)
        assert len({id(m) for m in pool.models}) == 1                             # one packed blob for every chain
        for _ in range(2):
            got = pool.map(batches, S=S, want_attn=True)
            torch.cuda.synchronize()
            for g, w in zip(got, want):
                assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])
    # the options the library ships with (persist_decode = 4; B = 4, T = 29 is inside the latency envelope): the pool's bits still do not depend on
    # how the list was cut - a one-batch group takes the grouped entry like any other (l2s_*_multi never takes the persistent form), so every
    # result equals the launch path's; the direct single-batch call is the latency form, another order of the same sums
    shipped = pc.shipped_model(synth_sd)
    for n_inflight, group in ((3, 1), (2, 3), (1, 8)):
        got = InflightPool(model=shipped, n_inflight=n_inflight, group=group).map(batches, S=S, want_attn=True)
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])
    if native.persist_available():
        direct = shipped.inference(*batches[0], S=S, want_attn=True)
        assert not torch.equal(direct[0], want[0][0]) and pc.maxdiff(direct[0], want[0][0]) < 5e-4
    assert InflightPool.chains_for(0, 8) == 1 and InflightPool.balanced_groups([], 8, 2) == []
    with pytest.raises(ValueError):
        InflightPool({k: v.cuda() for k, v in synth_sd.items()}, n_inflight=5)
    # host-resident batches prepared on the worker's stream (uint8 clips -> device normalise) and then grouped: same results
    from lip2speech_amd.datasets import PackedFrames
    gen = torch.Generator().manual_seed(5)
    raw = [(PackedFrames([torch.randint(0, 256, (T, 96, 96, 3), dtype=torch.uint8, generator=gen) for _ in range(B)]), b[1].cpu(), b[2].cpu()) for b in batches[:5]]
    prep = lambda b: (b[0].to_device(), b[1].cuda(), b[2].cuda())      # noqa: E731
    want_raw = [tuple(t.clone() for t in nm.inference(*prep(b), S=S, want_attn=True)) for b in raw]
    pool = InflightPool(model=nm, n_inflight=2, group=2)
    got = pool.map(raw, S=S, want_attn=True, prepare=prep, shape_of=lambda b: (B, 3, T, 96, 96))
    torch.cuda.synchronize()
    for g, w in zip(got, want_raw):
        assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])


@pytest.mark.gpu
@pytest.mark.parametrize("G,B", [(1, 32), (2, 32), (4, 32), (5, 32), (8, 32), (3, 17), (7, 20)])
def test_grouped_inference_is_bit_identical_per_batch(synth_sd, G, B):
    """`l2s_inference_multi`: G batches as rows of ONE launch chain (register-blocked 2x1 / 2x2 / 4x2 step kernels from 64 rows on) return,
    batch by batch, exactly what `l2s_inference` returns - mel, lengths and attention, bit for bit (B=32 is BASELINE.json's batch;
    17 exercises row tiles that straddle two batches)."""
    import parity_common as pc
    from lip2speech_amd import synth
    T, S = 29, 48
    batches = [(synth.synth_video(B, T, tag=f"grp{g}").cuda(), synth.synth_speaker_embedding(B, tag=f"grp{g}").cuda(),
                synth.synth_gumbel(B * 4, tag=f"grp{g}").cuda()) for g in range(G)]
    nm = pc.native_model(synth_sd)
    want = [tuple(t.clone() for t in nm.inference(*b, S=S, want_attn=True)) for b in batches]
    got = nm.inference_multi(batches, S=S, want_attn=True)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])
    # and the forced block shapes agree with each other (options are per model: this handle only).  The general block forms ("skinny_rc_jb" != 0)
    # cannot sum u = prenet + o in the loader, so LSTM layer 0 runs on [content | prenet | o | h0] (hoist_vproj = 1: another, equally valid, order of
    # additions): every form is compared on THAT layout - bit for bit among themselves - and the default layout against the single-batch calls above
    # first the split-bf16 LSTM blocks (default, "lstm_x3" = 2: eight waves) in every block shape, and their four-wave form: same bits as the default
    xown = pc.fresh_native_model(synth_sd, diag=True)
    for x3 in (2, 1):
        xown.set_option("lstm_x3", x3)
        for shape in (0, 11, 21, 22, 42):
            xown.set_option("skinny_rc", shape)
            alt = xown.inference_multi(batches, S=S, want_attn=True)
            for a, w in zip(alt, want):
                assert torch.equal(a[0], w[0]) and torch.equal(a[1], w[1]) and torch.equal(a[2], w[2]), (x3, shape)
    # the f32 block forms among themselves ("lstm_x3" = 0)
    own = pc.fresh_native_model(synth_sd, diag=True)
    own.set_option("lstm_x3", 0)
    own.set_option("hoist_vproj", 1)
    want = [tuple(t.clone() for t in own.inference(*b, S=S, want_attn=True)) for b in batches]
    assert all(pc.maxdiff(w[0], g[0]) < 1e-4 for w, g in zip(want, got))      # the layouts / matrix pipes agree to rounding
    for shape in (11, 21, 22, 42):
        own.set_option("skinny_rc", shape)
        for jb in (0, 2, 4, 15):
            own.set_option("skinny_rc_jb", jb)
            alt = own.inference_multi(batches, S=S, want_attn=True)
            for a, w in zip(alt, want):
                assert torch.equal(a[0], w[0]) and torch.equal(a[1], w[1]) and torch.equal(a[2], w[2]), (shape, jb)
    # the step's first phase as one flat grid of per-group block shapes (default from 128 rows on) against the uniform grid
    own.set_option("skinny_rc", 0)
    own.set_option("skinny_rc_jb", 0)
    own.set_option("skinny_flat", 0)
    alt = own.inference_multi(batches, S=S, want_attn=True)
    for a, w in zip(alt, want):
        assert torch.equal(a[0], w[0]) and torch.equal(a[1], w[1]) and torch.equal(a[2], w[2]), "skinny_flat=0"


@pytest.mark.gpu
def test_model_inference_pool_matches_model_inference():
    """`model.inference_pool(net)` (batches in flight) returns what `net.inference` returns, batch by batch."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from model.model import get_network
    from lip2speech_amd.model.model import inference_pool
    net = get_network("test").cuda()
    B, T = 3, 29
    batches = [(synth.synth_video(B, T, tag=f"mp{i}").cuda(), synth.synth_speaker_embedding(B, tag=f"mp{i}").cuda(),
                synth.synth_gumbel(B * 4, tag=f"mp{i}").cuda()) for i in range(4)]
    want = [net.inference(v, None, speaker_embedding=e, return_attention_map=True, gumbel_noise=g) for v, e, g in batches]
    got = inference_pool(net, n_inflight=2).map(batches, want_attn=True)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])


@pytest.mark.gpu
@pytest.mark.parametrize("G", [2, 8])
def test_grouped_forward_eval_is_bit_identical_per_batch(synth_sd, G):
    """`l2s_forward_eval_multi` at BASELINE.json's evaluate shape (B=32, T=29, S=77; evaluate.py:32-38 at tf_ratio=1): G loader batches as
    rows of ONE launch chain return, batch by batch, exactly what `l2s_forward_eval` returns on the batch alone - pre/post-net mel, stop
    logits, attention LOGITS and the content distribution, bit for bit - and the one-call form equals the five staged C-ABI calls."""
    import parity_common as pc
    B, T, S = 32, 29, 77
    batches = [(synth.synth_video(B, T, tag=f"fe{g}").cuda(), synth.synth_speaker_embedding(B, tag=f"fe{g}").cuda(),
                synth.synth_gumbel(B * 4, tag=f"fe{g}").cuda()) for g in range(G)]
    nm = pc.native_model(synth_sd)
    want = [tuple(t.clone() for t in nm.forward_eval(*b, S)) for b in batches]
    staged = nm.forward_eval_staged(*batches[0], S)
    assert all(torch.equal(a, b) for a, b in zip(staged, want[0])), "one-call forward_eval differs from the staged calls"
    got = nm.forward_eval_multi(batches, S)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert len(g) == 5 and all(torch.equal(a, b) for a, b in zip(g, w))
    # against the oracle: the first batch's first two clips (the whole batch would take the CPU a minute)
    with torch.no_grad():
        ref = orc.forward_eval(synth_sd, batches[0][0][:2].cpu(), batches[0][1][:2].cpu(), synth.synth_mels(2, S, tag="fe"), batches[0][2][:8].cpu())
    assert pc.maxdiff(got[0][1][:2], ref[1]) < MEL_TOL and pc.maxdiff(got[0][2][:2], ref[2][:, :, 0]) < 1e-4
    # teacher-forced steps (scheduled sampling at tf_ratio < 1): the group shares the mask, every batch brings its own frames
    mask = [1 if i % 3 == 1 else 0 for i in range(S)]
    bos = synth_sd["decoder.BOS"].expand(B, 1, -1)
    tb = [b + (torch.cat([bos, synth.synth_mels(B, S, tag=f"fet{g}").permute(0, 2, 1)[:, :S - 1]], dim=1).contiguous().cuda(),) for g, b in enumerate(batches[:3])]
    want_t = [tuple(t.clone() for t in nm.forward_eval(*b[:3], S, teacher=b[3], teacher_mask=mask)) for b in tb]
    got_t = nm.forward_eval_multi(tb, S, teacher_mask=mask)
    for g, w in zip(got_t, want_t):
        assert all(torch.equal(a, b) for a, b in zip(g, w))
    assert not torch.equal(want_t[0][0], want[0][0])                       # the forced frames do change the result


@pytest.mark.gpu
def test_forward_many_matches_forward_per_batch(synth_sd):
    """`Lip2Speech.forward_many` (evaluate.py's loop on the grouped path, `InflightPool.imap`): 11 loader batches of B=32, S=77 run as
    chains of 8 + 3 (ragged last group) on two streams and come back in order, each bit-identical to `net(..., tf_ratio=1)`; a batch
    with another S closes the running group; host-resident inputs are staged by the pool."""
    from model.model import get_network
    B, T, S = 32, 29, 77
    net = get_network("test")
    net.load_state_dict(synth_sd, strict=True)
    net = net.cuda()
    n = 11
    vids = [synth.synth_video(B, T, tag=f"fm{i % 4}") for i in range(n)]                 # host tensors: the pool copies them
    embs = [synth.synth_speaker_embedding(B, tag=f"fm{i}").cuda() for i in range(n)]
    gums = [synth.synth_gumbel(B * 4, tag=f"fm{i}").cuda() for i in range(n)]
    mels = [torch.zeros(B, 80, S if i != 5 else 40) for i in range(n)]                   # batch 5: S = 40 -> groups 5 + 1 + 5
    lens = torch.full((B,), T)
    calls = [(vids[i], None, None, mels[i], lens, None, None, 1, {"speaker_embedding": embs[i], "gumbel_noise": gums[i]}) for i in range(n)]
    with torch.no_grad():
        want = [[t.clone() if isinstance(t, torch.Tensor) else t for t in net(vids[i].cuda(), None, None, mels[i].cuda(), lens, None, None, 1,
                                                                             speaker_embedding=embs[i], gumbel_noise=gums[i])] for i in range(n)]
    nm = net.native_model()
    nm.calls.clear()
    got = list(net.forward_many(iter(calls), group=8, n_inflight=2))
    torch.cuda.synchronize()
    assert len(got) == n
    for g, w in zip(got, want):
        assert len(g) == 7 and g[6] is lens
        assert all(torch.equal(a, b) for a, b in zip(g[:6], w[:6]))
    assert nm.calls["l2s_forward_eval_multi"] == 2 and nm.calls["l2s_forward_eval"] == 1
    stats = net.pool(8, 2).stats
    assert stats["forward_batches"] == n and stats["max_group"] == 5
    # no S change: 8 + 3
    calls2 = [c for i, c in enumerate(calls) if i != 5] + [calls[0]]
    nm.calls.clear()
    net.pool(8, 2).stats.clear()
    got2 = list(net.forward_many(calls2, group=8, n_inflight=2))
    assert net.pool(8, 2).stats["max_group"] == 8 and nm.calls["l2s_forward_eval_multi"] == 2
    for g, w in zip(got2, [w for i, w in enumerate(want) if i != 5] + [want[0]]):
        assert all(torch.equal(a, b) for a, b in zip(g[:6], w[:6]))
    # inference_many: the same for demo.py's loop (S = 300 cut to a short clip count here)
    icalls = [(vids[i][:4], None, embs[i][:4], True, gums[i][:16]) for i in range(5)]
    with torch.no_grad():
        iwant = [net.inference(vids[i][:4].cuda(), None, embs[i][:4], True, gums[i][:16]) for i in range(5)]
    igot = list(net.inference_many(icalls, group=4, n_inflight=2))
    for g, w in zip(igot, iwant):
        assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) and torch.equal(g[2], w[2])


@pytest.mark.gpu
def test_stop_bookkeeping_matches_reference_golden(synth_sd, persist_default):
    """decoder.py:429-435 pinned against the reference where it is not trivial (tests/golden/make_stop_goldens.py: the checkpoint differs only
    in the stop layer; the reference's first crossings spread over 13..286 and three clips of the B=32 batch never stop -> 300; at B=2 one clip
    stops at 183, the other never): int64 equality of `output_lengths` through `l2s_inference` AND `l2s_inference_multi`, the stop logits
    of the staged calls, and - B=2 - the FULL (2,300,29) post-softmax attention tensor."""
    import parity_common as pc
    cases = {}
    for name, B, vtag, etag in (("stop_lrw_b2.npz", 2, "video-lrw2", "spk-lrw2"), ("stop_lrw_b32.npz", 32, "bench", "bench")):
        g = pc.golden(name)
        sd = dict(synth_sd)
        sd["decoder.stop_token_layer.linear_layer.weight"] = g["stop_weight"]
        sd["decoder.stop_token_layer.linear_layer.bias"] = g["stop_bias"]
        own = pc.fresh_native_model(sd)
        args = (synth.synth_video(B, 29, tag=vtag).cuda(), synth.synth_speaker_embedding(B, tag=etag).cuda(), g["gumbel"].cuda())
        mel_post, lengths, attn = own.inference(*args, S=300, want_attn=True)
        assert lengths.dtype == torch.int64 and torch.equal(lengths.cpu(), g["output_lengths"]), (name, lengths.cpu(), g["output_lengths"])
        cases[name] = (own, args, g, mel_post.clone(), attn.clone())
    # B=2: every element of the attention tensor and the mel against the reference
    own, args, g, mel_post, attn = cases["stop_lrw_b2.npz"]
    assert tuple(attn.shape) == (2, 300, 29) and pc.maxdiff(attn, g["attn"]) < MEL_TOL
    assert pc.maxdiff(mel_post, pc.golden("inference_lrw_b2.npz")["mel_post"]) < MEL_TOL
    lens2 = {int(x) for x in g["output_lengths"]}
    assert 300 in lens2 and any(10 < x < 300 for x in lens2)
    # B=32 through the grouped entry: the golden batch as group member 0 and 2 (other clips in between), lengths per batch
    own, args, g, mel_post, attn = cases["stop_lrw_b32.npz"]
    lens = g["output_lengths"]
    assert len({int(x) for x in lens if int(x) > 10}) >= 5 and int((lens == 300).sum()) >= 2
    other = (synth.synth_video(32, 29, tag="grp1").cuda(), synth.synth_speaker_embedding(32, tag="grp1").cuda(), synth.synth_gumbel(32 * 4, tag="grp1").cuda())
    want_other = own.inference(*other, S=300)[1].clone()
    got = own.inference_multi([args, other, args], S=300)
    assert torch.equal(got[0][1].cpu(), lens) and torch.equal(got[2][1].cpu(), lens) and torch.equal(got[1][1], want_other)
    assert got[0][1].dtype == torch.int64 and torch.equal(got[0][0], mel_post)
    # the stop logits themselves (staged calls return them; forward_eval runs the same step with teacher = none for S = 300)
    out = own.forward_eval(args[0], args[1], args[2], 300)
    assert pc.maxdiff(out[2].reshape(32, 300), g["stop_logits"]) < 1e-3


@pytest.mark.gpu
def test_persistent_decode_matches_reference_golden(synth_sd, nm):
    """Option "persist_decode" (pdecode.hip: the whole free-running loop as ONE weight-stationary launch for one or two clips; demo.py:60-90 runs one clip,
    BASELINE config 1 two) against the REFERENCE's B=2 golden - mel, lengths, attention argmax, and the full (2,300,29) attention tensor of the
    stop-layer golden with its non-trivial output lengths - and against the launch path (another order of the same fp32 sums)."""
    import parity_common as pc
    g, video, emb = pc.lrw2_inputs()
    own = pc.fresh_native_model(synth_sd, persist_decode=8)
    args = (video.cuda(), emb.cuda(), g["gumbel"].cuda())
    for _ in range(2):                                     # twice: the exchange buffer is re-zeroed per call
        mel_post, lengths, attn = own.inference(*args, S=300, want_attn=True)
    assert torch.isfinite(mel_post).all()
    assert pc.maxdiff(mel_post, g["mel_post"]) < MEL_TOL
    assert torch.equal(lengths.cpu(), g["output_lengths"])
    amax, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    assert torch.equal(amax[sure], g["attn_argmax"][sure])
    ref = nm.inference(*args, S=300, want_attn=True)
    assert pc.maxdiff(mel_post, ref[0]) < 5e-4 and pc.maxdiff(attn, ref[2]) < 5e-4
    # the staged entry point takes the same route (l2s_decode_steps); pre-softmax attention LOGITS (what forward() returns; |values| in the thousands,
    # fp64-accumulated on the launch path, fp32 here), the pre-post-net mel and the stop logits against the launch path
    feat = nm.encoder_fwd(args[0]); vis = native.build_visual(feat, args[1]); state, _ = nm.decoder_prologue(vis, args[1], args[2])
    pa = own.decode_steps(state, 2, 29, 60, attn_logits=True)
    pb = nm.decode_steps(state, 2, 29, 60, attn_logits=True)
    assert torch.equal(pa[0], pb[0]) != native.persist_available() and pc.maxdiff(pa[0], pb[0]) < 2e-4 and pc.maxdiff(pa[1], pb[1]) < 2e-4
    assert pc.maxdiff(pa[2], pb[2]) / pb[2].abs().max().item() < 1e-5
    gs = pc.golden("stop_lrw_b2.npz")
    sd = dict(synth_sd)
    sd["decoder.stop_token_layer.linear_layer.weight"] = gs["stop_weight"]
    sd["decoder.stop_token_layer.linear_layer.bias"] = gs["stop_bias"]
    own2 = pc.fresh_native_model(sd, persist_decode=8)
    sargs = (synth.synth_video(2, 29, tag="video-lrw2").cuda(), synth.synth_speaker_embedding(2, tag="spk-lrw2").cuda(), gs["gumbel"].cuda())
    mel_post, lengths, attn = own2.inference(*sargs, S=300, want_attn=True)
    assert torch.equal(lengths.cpu(), gs["output_lengths"]) and pc.maxdiff(attn, gs["attn"]) < MEL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,HW,S", [(1, 29, 96, 300), (2, 13, 88, 25), (2, 32, 88, 12), (1, 7, 96, 1), (2, 8, 96, 9), (1, 32, 96, 5), (3, 29, 96, 40), (4, 13, 88, 25)])
def test_persistent_decode_shapes_against_launch_path(synth_sd, nm, B, T, HW, S):
    """One to four clips (two per launch: three and four clips are two launches), T up to the 32 frames whose keys / values a workgroup holds, S = 1: the
    persistent loop against the launch-per-phase loop on the same inputs (another order of the same fp32 sums); calls outside its limits (5 clips, 33
    frames) take the launch path: same bits."""
    import parity_common as pc
    own = pc.fresh_native_model(synth_sd, persist_decode=8)
    video = synth.synth_video(B, T, H=HW, W=HW, tag=f"pd{B}").cuda()
    emb = synth.synth_speaker_embedding(B, tag=f"pd{B}").cuda()
    gum = synth.synth_gumbel(B * native.min_T(T), tag=f"pd{B}").cuda()
    a = own.inference(video, emb, gum, S=S, want_attn=True)
    b = nm.inference(video, emb, gum, S=S, want_attn=True)
    assert torch.isfinite(a[0]).all()
    assert torch.equal(a[0], b[0]) != native.persist_available()      # it did take the other route - wherever the device allows it (l2s_persist_available)
    assert native.persist_timeouts() == 0
    assert pc.maxdiff(a[0], b[0]) < 5e-4 and torch.equal(a[1], b[1]) and pc.maxdiff(a[2], b[2]) < 5e-4
    if B == 1 and S == 300:
        v3 = synth.synth_video(5, 8, tag="pd5").cuda(); e3 = synth.synth_speaker_embedding(5, tag="pd5").cuda(); g3 = synth.synth_gumbel(5 * native.min_T(8), tag="pd5").cuda()
        assert torch.equal(own.inference(v3, e3, g3, S=5)[0], nm.inference(v3, e3, g3, S=5)[0])          # > 4 clips: the launch path, same bits
        v33 = synth.synth_video(2, 33, tag="pd33").cuda(); e33 = synth.synth_speaker_embedding(2, tag="pd33").cuda(); g33 = synth.synth_gumbel(2 * native.min_T(33), tag="pd33").cuda()
        assert torch.equal(own.inference(v33, e33, g33, S=5)[0], nm.inference(v33, e33, g33, S=5)[0])    # > 32 frames: the launch path
