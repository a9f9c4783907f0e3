#!/usr/bin/env python
"""Stage-by-stage deviation table of the HIP path against the oracle / committed goldens.
Run on the GPU box:  python tools/gpu_diag.py   (prints one line per checked tensor)."""
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from lip2speech_amd import native, synth          # noqa: E402
from oracle import l2s_oracle as orc               # noqa: E402
import parity_common as pc                         # noqa: E402


def line(tag, d, tol):
    print(f"{'OK ' if d < tol else 'BAD'} {tag:<34s} max|d| = {d:.3e}   (tol {tol:g})", flush=True)


def section(fn):
    try:
        fn()
    except Exception:
        print(f"EXC in {fn.__name__}:\n{traceback.format_exc()}", flush=True)


def gemm_cases():
    torch.manual_seed(0)
    for (M, N, K, act) in [(64, 64, 32, 0), (100, 70, 36, 1), (928, 512, 1024, 2), (37, 501, 256, 3), (130, 58, 58, 1), (9, 80, 2560, 0)]:
        A = torch.randn(M, K).cuda(); W = (torch.randn(N, K) / K ** 0.5).cuda()
        sc = (torch.rand(N) + 0.5).cuda(); sh = torch.randn(N).cuda(); aw = (torch.rand(N) + 0.5).cuda()
        C = native.op_gemm(A, W, sc, sh, aw, act)
        ref = (A.double() @ W.double().t()) * sc.double() + sh.double()
        if act == 1: ref = ref.relu()
        if act == 2: ref = ref * torch.sigmoid(ref)
        if act == 3: ref = torch.sin(ref) * aw.double()
        line(f"gemm M{M} N{N} K{K} act{act}", pc.maxdiff(C, ref), 2e-5)


def conv1d_cases():
    torch.manual_seed(1)
    for (B, T, Ci, Co, k, st, pad) in [(2, 29, 512, 512, 11, 1, 5), (2, 29, 512, 512, 7, 7, 0), (3, 40, 80, 512, 5, 1, 2), (2, 75, 512, 80, 5, 1, 2), (2, 29, 512, 512, 3, 3, 0)]:
        X = torch.randn(B, T, Ci).cuda(); Wt = (torch.randn(Co, Ci, k) / (Ci * k) ** 0.5)
        Wp = Wt.permute(0, 2, 1).reshape(Co, k * Ci).contiguous().cuda()
        out = native.op_conv1d(X, Wp, taps=k, stride=st, pad=pad)
        ref = torch.nn.functional.conv1d(X.cpu().double().permute(0, 2, 1), Wt.double(), stride=st, padding=pad).permute(0, 2, 1)
        line(f"conv1d T{T} {Ci}->{Co} k{k} s{st} p{pad}", pc.maxdiff(out, ref), 2e-5)


def encoder_stages():
    sd = synth.synth_state_dict()
    nm = pc.native_model(sd)
    g, video, emb = pc.lrw2_inputs()
    # frontend on a short clip against the oracle (CPU)
    v = video[:1, :, :4].contiguous()
    fr = nm.op_frontend(v.cuda())
    ref = orc.frontend3d(v, sd).permute(0, 2, 3, 1)
    line("frontend3d 96x96 (1 clip, 4 frames)", pc.maxdiff(fr, ref), 2e-5)
    v88 = synth.synth_video(1, 3, 88, 88, tag="v88")
    fr = nm.op_frontend(v88.cuda())
    ref = orc.frontend3d(v88, sd).permute(0, 2, 3, 1)
    line("frontend3d 88x88", pc.maxdiff(fr, ref), 2e-5)
    full = nm.op_frontend(video.cuda())
    frames = g["frames"].tolist()
    line("frontend3d vs golden taps", pc.maxdiff(full[frames], g["oracle_frontend"].permute(0, 2, 3, 1)), 2e-5)
    for fuse in (0, 1):
        nm.set_option("fuse_trunk", fuse)
        feat = nm.encoder_fwd(video.cuda())
        line(f"fuse_trunk={fuse} encoder feat vs reference golden", pc.maxdiff(feat, g["feat"]), 1e-5)
        v32 = synth.synth_video(32, 29, tag="bench").cuda()
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time(); nm.encoder_fwd(v32); torch.cuda.synchronize()
            print(f"   fuse_trunk={fuse} encoder B=32: {(time.time() - t0) * 1e3:.2f} ms")
    feat88 = nm.encoder_fwd(v88.cuda())
    line("encoder feat 88x88 vs oracle", pc.maxdiff(feat88, orc.encoder_forward(sd, v88)), 1e-5)


def decoder_stages():
    sd = synth.synth_state_dict()
    nm = pc.native_model(sd)
    g, video, emb = pc.lrw2_inputs()
    B, T, S = 2, 29, 300
    vis = orc.build_visual(g["feat"], emb).cuda()
    state, dis = nm.decoder_prologue(vis, emb.cuda(), g["gumbel"].cuda())
    sf = lambda f, shape: native.state_field(state, B, T, f, shape)
    line("prologue enc", pc.maxdiff(sf(native.ST_ENC, (B, T, 512)), g["oracle_enc"]), 2e-5)
    line("prologue k", pc.maxdiff(sf(native.ST_K, (B, T, 512)), g["oracle_k"].permute(0, 2, 1)), 5e-5)
    line("prologue v", pc.maxdiff(sf(native.ST_V, (B, T, 512)), g["oracle_v"]), 5e-5)
    line("prologue content key", pc.maxdiff(sf(native.ST_CKEY, (B, 4, 256)), g["oracle_key"].permute(0, 2, 1)), 2e-5)
    line("prologue content value", pc.maxdiff(sf(native.ST_CVAL, (B, 4, 256)), g["oracle_value"]), 1e-4)
    line("prologue encoder_cell", pc.maxdiff(sf(native.ST_ECELL, (B, 512)), g["oracle_encoder_cell"]), 2e-5)
    hfrag = sf(native.ST_H, (2, 16 * 512))
    line("prologue hidden[0]", pc.maxdiff(pc.unfrag(hfrag[0], B, 512), g["oracle_hidden"][0]), 2e-5)
    line("prologue hidden[1]", pc.maxdiff(pc.unfrag(hfrag[1], B, 512), g["oracle_hidden"][1]), 2e-5)
    fg = pc.golden("forward_lrw_b2_s77.npz")
    line("prologue content_dis", pc.maxdiff(dis, fg["content_dis"]), 1e-6)
    for fold, graph in ((0, 0), (1, 0), (0, 1), (1, 1)):
        nm.set_option("fold_step_weights", fold); nm.set_option("use_graph", graph)
        for Sx in (3, 77, 300):
            mel, stop, attn = nm.decode_steps(state, B, T, Sx, want_attn=True)
            line(f"fold{fold} graph{graph} decode S={Sx} mel_pre", pc.maxdiff(mel.permute(0, 2, 1), g["oracle_mel_pre"][:, :, :Sx]), 1e-3)
            line(f"fold{fold} graph{graph} decode S={Sx} stop", pc.maxdiff(stop, g["oracle_stop"][:, :Sx]), 1e-3)
        post, _ = nm.postnet(mel)
        line(f"fold{fold} graph{graph} mel_post vs reference", pc.maxdiff(post, g["mel_post"]), 1e-3)
        am, _ = pc.top2(attn.cpu())
        sure0 = g["attn_margin"] > 1e-4
        print("   attention argmax mismatches:", int((am[sure0] != g["attn_argmax"][sure0]).sum()))
    am, _ = pc.top2(attn.cpu())
    sure = g["attn_margin"] > 1e-4
    print("   attention argmax mismatches (margin>1e-4):", int((am[sure] != g["attn_argmax"][sure]).sum()), "of", int(sure.sum()))
    line("attention rows", pc.maxdiff(attn[:, ::50], g["attn_rows"]), 1e-3)
    post, _ = nm.postnet(mel)
    line("mel_post S=300 vs reference", pc.maxdiff(post, g["mel_post"]), 1e-3)
    post_only, _ = nm.postnet(g["oracle_mel_pre"].permute(0, 2, 1).contiguous().cuda())
    line("postnet alone (oracle mel in)", pc.maxdiff(post_only, g["mel_post"]), 1e-4)
    lens = native.output_lengths(stop)
    print("   output_lengths", lens.tolist(), "golden", g["output_lengths"].tolist())
    nm.set_option("fold_step_weights", 1); nm.set_option("use_graph", 0)
    for ov in (0, 1, 1):
        nm.set_option("overlap_postnet", ov)
        t0 = time.time()
        mp, ln, at = nm.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=300, want_attn=True)
        torch.cuda.synchronize()
        line(f"overlap_postnet={ov} l2s_inference mel_post", pc.maxdiff(mp, g["mel_post"]), 1e-3)
        print(f"   l2s_inference B=2 wall {time.time() - t0:.3f}s lengths {ln.tolist()}")
    for Sx in (5, 13, 70, 130):
        nm.set_option("overlap_postnet", 0)
        a, _, _ = nm.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=Sx)
        nm.set_option("overlap_postnet", 1)
        b, _, _ = nm.inference(video.cuda(), emb.cuda(), g["gumbel"].cuda(), S=Sx)
        line(f"overlap vs sequential post-net, S={Sx}", pc.maxdiff(a, b), 1e-12)


def timing():
    nm = pc.native_model()
    B, T, S = 32, 29, 300
    video = synth.synth_video(B, T, tag="bench").cuda()
    emb = synth.synth_speaker_embedding(B, tag="bench").cuda()
    gum = synth.synth_gumbel(B * 4, tag="bench").cuda()
    for fold, graph, ov in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 0, 1)):
        nm.set_option("fold_step_weights", fold); nm.set_option("use_graph", graph); nm.set_option("overlap_postnet", ov)
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            nm.inference(video, emb, gum, S=S)
            torch.cuda.synchronize(); dt = time.time() - t0
            print(f"   fold{fold} graph{graph} overlap{ov} B=32 inference iter {it}: {dt * 1e3:.1f} ms -> {B * S / dt:.0f} mel-frames/s", flush=True)
    native.profile_enable(True); native.profile_reset()
    nm.inference(video, emb, gum, S=S)
    torch.cuda.synchronize()
    rows = sorted(native.profile_read(), key=lambda r: -r[2])
    native.profile_enable(False)
    for name, n, ms in rows:
        print(f"   {name:<34s} launches {n:5d}  total {ms:8.3f} ms  avg {ms / max(n, 1) * 1e3:8.1f} us")


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0))
    for fn in (gemm_cases, conv1d_cases, encoder_stages, decoder_stages, timing):
        print(f"== {fn.__name__}", flush=True)
        section(fn)
