"""Training-side primitives: loss terms, fused clip + AdamW(amsgrad), bucketed gradient all-reduce."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _allreduce_worker(rank, size, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    from lip2speech_amd.training import GradAllReducer
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = GradAllReducer(g, bucket_bytes=1024)          # 256 floats per bucket -> 4 buckets
    assert len(red.buckets) == 4
    assert red.buckets_covering(600) == 2 and red.buckets_covering(1000) == 4 and red.buckets_covering(100) == 0
    red.start(0, red.buckets_covering(600))             # the decoder group's buckets first, as soon as its gradients are final
    red.start(red.buckets_covering(600), None)
    mul = red.wait()
    ret[rank] = (g.tolist(), mul)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_allreduce_worker, args=(2, 29517, ret), nprocs=2, join=True)
        want = (torch.arange(1000, dtype=torch.float32) * 3).tolist()
        for r in range(2):
            vals, mul = ret[r]
            assert vals == want and mul == 0.5


@pytest.mark.gpu
def test_loss_terms_match_reference_formulas():
    from lip2speech_amd.training import loss_terms
    torch.manual_seed(0)
    B, S, R = 4, 77, 16
    mel, post, tgt = torch.randn(B, 80, S), torch.randn(B, 80, S), torch.randn(B, 80, S) - 4
    stop, gate = torch.randn(B, S, 1) * 3, (torch.rand(B, S) > 0.7).float()
    dis = torch.softmax(torch.randn(R, 501) * 2, dim=-1)
    leaves = [t.clone().double().requires_grad_(True) for t in (mel, post, stop, dis)]
    m, p, s, q = leaves
    # train_utils/losses.py:69-77
    kld = torch.sum(q * torch.log(q * 501 + 1e-20), dim=-1).mean()
    mel_loss = torch.nn.functional.mse_loss(m, tgt.double())
    post_loss = 10 * torch.nn.functional.mse_loss(p, tgt.double())
    gate_loss = torch.nn.functional.binary_cross_entropy_with_logits(s.view(-1, 1), gate.double().view(-1, 1))
    total = kld + mel_loss + post_loss + gate_loss
    total.backward()
    out, grads = loss_terms(mel.cuda(), post.cuda(), stop.cuda(), dis.cuda(), tgt.cuda(), gate.cuda())
    want = torch.stack([mel_loss, post_loss, gate_loss, kld, total]).float()
    assert (out.cpu() - want).abs().max() < 1e-5 * want.abs().max()
    for key, leaf in zip(("mel", "mel_post", "stop", "content_dis"), leaves):
        ref = leaf.grad.float().reshape(grads[key].shape)
        assert (grads[key].cpu() - ref).abs().max() <= 1e-6 * max(1.0, ref.abs().max().item()), key


@pytest.mark.gpu
def test_fused_adamw_amsgrad_clip_matches_torch():
    from lip2speech_amd.training import AdamWAmsgrad, FlatBuffer
    torch.manual_seed(1)
    shapes = [(37, 19), (501,), (8, 4, 3), (1,)]
    ref_params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    my_params = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_params]
    opt_ref = torch.optim.AdamW([{"params": ref_params[:2]}, {"params": ref_params[2:]}], lr=1e-2, weight_decay=1e-2, amsgrad=True)
    flat = FlatBuffer([my_params[:2], my_params[2:]])
    opt = AdamWAmsgrad(flat, lr=1e-2, weight_decay=1e-2)
    assert all(p.data.data_ptr() >= flat.data.data_ptr() for p in my_params)
    for step in range(6):
        scale = 10.0 if step % 2 == 0 else 0.01                 # alternate clipped / unclipped steps
        grads = [torch.randn(s) * scale for s in shapes]
        for p, g in zip(ref_params, grads):
            p.grad = g.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
        opt_ref.step()
        for p, g in zip(my_params, grads):
            p.grad.copy_(g.cuda())
        norm = opt.step(max_norm=1.0)
        assert abs(norm.item() - norm_ref.item()) < 1e-5 * norm_ref.item()
        for p, q in zip(my_params, ref_params):
            assert (p.detach().cpu() - q.detach()).abs().max() < 2e-6, step
    # data-parallel form: gradient summed over 4 ranks, averaged inside the update
    g = torch.randn(flat.numel).cuda()
    a, b = flat.data.clone(), None
    flat.grad.copy_(g * 4)
    opt.step(max_norm=1.0, grad_mul=0.25)
    b = flat.data.clone()
    assert (a - b).abs().max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,Ci,Co,k,st,pad", [(2, 29, 512, 512, 11, 1, 5), (3, 40, 80, 512, 5, 1, 2), (2, 75, 512, 80, 5, 1, 2),
                                                (3, 144, 58, 58, 1, 1, 0), (2, 100, 24, 58, 1, 1, 0), (1, 333, 58, 116, 1, 1, 0),
                                                 (2, 29, 512, 512, 7, 7, 0), (4, 1, 1024, 512, 1, 1, 0), (2, 29, 2560, 256, 1, 1, 0)])
def test_conv1d_backward_gemms(B, T, Ci, Co, k, st, pad):
    """dX (flipped-tap implicit GEMM) and dW (reduction over rows) of the Conv1d/Linear layers vs autograd in fp64."""
    from lip2speech_amd import native
    torch.manual_seed(T + k + Co)
    X = torch.randn(B, T, Ci)
    Wt = torch.randn(Co, Ci, k) / (Ci * k) ** 0.5
    Wp = Wt.permute(0, 2, 1).reshape(Co, k * Ci).contiguous()
    Xd, Wd = X.double().requires_grad_(True), Wt.double().requires_grad_(True)
    out = torch.nn.functional.conv1d(Xd.permute(0, 2, 1), Wd, stride=st, padding=pad).permute(0, 2, 1)
    dZ = torch.randn(out.shape)
    out.backward(dZ.double())
    dX, dW = native.op_conv1d_bwd(dZ.cuda(), X.cuda(), Wp.cuda(), taps=k, stride=st, pad=pad, want_dx=(st == 1))
    dW_ref = Wd.grad.permute(0, 2, 1).reshape(Co, k * Ci)
    assert (dW.cpu().double() - dW_ref).abs().max() < 2e-5 * max(1.0, dW_ref.abs().max().item())
    if st == 1:
        assert (dX.cpu().double() - Xd.grad).abs().max() < 2e-5 * max(1.0, Xd.grad.abs().max().item())


@pytest.mark.gpu
def test_postnet_backward_matches_autograd(synth_sd):
    """Stage 1 of the model's training path: post-net forward-with-tape and backward (input and every parameter gradient)
    against autograd through the oracle's post-net in fp64."""
    import parity_common as pc
    from oracle import l2s_oracle as orc
    B, S = 3, 41
    torch.manual_seed(3)
    mel = torch.randn(B, S, 80)
    dpost = torch.randn(B, 80, S)
    keys = [k for k in synth_sd if k.startswith("decoder.postnet.") and synth_sd[k].is_floating_point()]
    sd64 = {k: synth_sd[k].double().requires_grad_(not k.endswith(("running_mean", "running_var"))) for k in keys}
    mel_cf = mel.double().permute(0, 2, 1).contiguous().requires_grad_(True)
    out = orc.postnet(sd64, mel_cf) + mel_cf
    out.backward(dpost.double())
    nm = pc.native_model(synth_sd)
    params = {k: synth_sd[k].cuda() for k in keys}
    grads = {k: torch.zeros_like(v) for k, v in params.items() if not k.endswith(("running_mean", "running_var"))}
    nm.train_bind(params, grads)
    got_out, dmel = nm.train_postnet(mel.cuda(), dpost.cuda())
    assert pc.maxdiff(got_out, out) < 1e-4
    ref_dmel = mel_cf.grad.permute(0, 2, 1)
    assert pc.maxdiff(dmel, ref_dmel) < 2e-4 * max(1.0, ref_dmel.abs().max().item())
    for k, g in grads.items():
        ref = sd64[k].grad
        assert ref is not None, k
        assert pc.maxdiff(g, ref.reshape(g.shape)) < 3e-4 * max(1.0, ref.abs().max().item()), k


STEP_KEYS = ["fc_out.linear_layer.weight", "fc_out.linear_layer.bias", "stop_token_layer.linear_layer.weight", "stop_token_layer.linear_layer.bias",
             "decoder_rnn.weight_ih_l0", "decoder_rnn.weight_hh_l0", "decoder_rnn.bias_ih_l0", "decoder_rnn.bias_hh_l0",
             "decoder_rnn.weight_ih_l1", "decoder_rnn.weight_hh_l1", "decoder_rnn.bias_ih_l1", "decoder_rnn.bias_hh_l1",
             "attention_proj.linear_layer.weight", "attention_proj.linear_layer.bias", "Q.0.linear_layer.weight", "Q.0.linear_layer.bias", "Q.1.w",
             "content.Q.0.weight", "content.Q.0.bias", "prenet.0.linear_layer.weight", "prenet.0.linear_layer.bias", "prenet.1.w",
             "prenet.3.linear_layer.weight", "prenet.3.linear_layer.bias", "prenet.4.w", "BOS", "temperature", "content.temperature"]


@pytest.mark.gpu
@pytest.mark.parametrize("S,forced", [(9, False), (12, True)])
def test_decode_loop_bptt_matches_autograd(synth_sd, S, forced):
    """Stage 2 of the training path: the S-step loop with a tape and its back-propagation through time against autograd through
    the oracle's decode loop in fp64 - every step parameter and every state tensor the prologue produced."""
    import parity_common as pc
    from lip2speech_amd import native, synth
    from oracle import l2s_oracle as orc
    g, _, emb = pc.lrw2_inputs()
    B, T = 2, 29
    nm = pc.native_model(synth_sd)
    vis = native.build_visual(g["feat"].cuda(), emb.cuda())
    state, _ = nm.decoder_prologue(vis, emb.cuda(), g["gumbel"].cuda())
    torch.manual_seed(S)
    Gm, Gs = torch.randn(B, S, 80), torch.randn(B, S)
    mels = synth.synth_mels(B, S, tag="mel-lrw2")
    mask = None
    teacher = None
    if forced:
        mask = torch.zeros(S, dtype=torch.bool)
        mask[[0, 3, 4, 9]] = True
        teacher = torch.cat([synth_sd["decoder.BOS"].expand(B, 1, -1), mels.permute(0, 2, 1)[:, :S - 1]], dim=1).contiguous()
    keys = ["decoder." + k for k in STEP_KEYS] + ["decoder.positional_encodings.pos_table"]
    sd64 = {k: synth_sd[k].double().requires_grad_(k != "decoder.positional_encodings.pos_table") for k in keys}
    st = {"k": g["oracle_k"], "v": g["oracle_v"], "key": g["oracle_key"], "value": g["oracle_value"], "hidden": g["oracle_hidden"],
          "encoder_cell": g["oracle_encoder_cell"]}
    st64 = {k: v.double().requires_grad_(True) for k, v in st.items()}
    # the reference builds teacher_input = cat(BOS, mels) from the BOS parameter itself (decoder.py:349), so its gradient reaches BOS
    teacher64 = torch.cat([sd64["decoder.BOS"].expand(B, 1, -1), mels.double().permute(0, 2, 1)[:, :S - 1]], dim=1) if forced else None
    mel_o, stop_o, logit_o = orc.decode_loop(sd64, st64, S, teacher=teacher64, teacher_mask=mask, return_logits=True)
    ((mel_o * Gm.double()).sum() + (stop_o * Gs.double()).sum()).backward()
    params = {k: synth_sd[k].cuda() for k in keys if k != "decoder.positional_encodings.pos_table"}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    nm.train_bind(params, grads)
    (mel, stop, logits), sg = nm.train_steps(state, B, T, S, Gm.cuda(), Gs.cuda(), teacher=teacher.cuda() if forced else None,
                                             teacher_mask=mask.numpy() if forced else None)
    assert pc.maxdiff(mel, mel_o) < 1e-4 and pc.maxdiff(stop, stop_o) < 1e-4
    assert pc.maxdiff(logits, logit_o) / logit_o.abs().max().item() < 1e-5

    def close(name, got, ref, rel=2e-3):
        ref = ref.reshape(got.shape)
        scale = max(1e-6, ref.abs().max().item())
        err = pc.maxdiff(got, ref) / scale
        assert err < rel, f"{name}: relative error {err:.2e} (scale {scale:.2e})"

    close("dk", sg["dk"], st64["k"].grad.permute(0, 2, 1))
    close("dv", sg["dv"], st64["v"].grad)
    close("dckey", sg["dckey"], st64["key"].grad.permute(0, 2, 1))
    close("dcval", sg["dcval"], st64["value"].grad)
    close("dh_init", sg["dh_init"], st64["hidden"].grad)
    close("de_c", sg["de_c"], st64["encoder_cell"].grad)
    for k in grads:
        assert sd64[k].grad is not None, k
        close(k, grads[k], sd64[k].grad)


PROLOGUE_PREFIXES = ("residual_bottleneck.", "encoder_site.", "attention_site.", "encoder_rnn.", "E_C.", "encoder_proj.", "K.", "V.", "content.")
PROLOGUE_SKIP = ("content.Q.0.", "content.temperature")          # used by the loop, not by the prologue


@pytest.mark.gpu
@pytest.mark.parametrize("T", [29, 40])
def test_prologue_backward_matches_autograd(synth_sd, T):
    """Stage 3 of the training path: decoder prologue (site embeddings, BiLSTM, MultiHop K/V, Content.encode with the Gumbel
    soft-max) forward-with-tape and backward against autograd through the oracle's prologue in fp64: the gradient wrt the visual
    features and every prologue parameter."""
    import parity_common as pc
    from lip2speech_amd import native, synth
    from oracle import l2s_oracle as orc
    B = 2
    m = native.min_T(T)
    torch.manual_seed(T)
    feat = torch.nn.functional.normalize(torch.randn(B, T, 768), dim=-1)
    emb = synth.synth_speaker_embedding(B, tag="pro-train")
    gumbel = synth.synth_gumbel(B * m, tag="pro-train")
    vis = orc.build_visual(feat, emb).contiguous()
    dec = [k for k in synth_sd if k.startswith("decoder.") and synth_sd[k].is_floating_point() and not k.startswith("decoder.postnet.")]
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "pos_table"))
    pro = [k for k in dec if k[len("decoder."):].startswith(PROLOGUE_PREFIXES) and not k[len("decoder."):].startswith(PROLOGUE_SKIP) and not is_buf(k)]
    sd64 = {k: synth_sd[k].double().requires_grad_(k in pro) for k in dec}
    vis64 = vis.double().requires_grad_(True)
    st = orc.decoder_prologue(sd64, vis64, emb.double(), gumbel.double())
    names = ["k", "v", "key", "value", "hidden", "encoder_cell", "content_dis"]
    cot = {n: torch.randn(st[n].shape, dtype=torch.float64) for n in names}
    cot["content_dis"] *= 50.0            # the distribution is nearly uniform over 501 words; give its path a visible gradient
    sum((st[n] * cot[n]).sum() for n in names).backward()

    nm = pc.native_model(synth_sd)
    params = {k: synth_sd[k].cuda() for k in dec if not is_buf(k)}
    grads = {k: torch.zeros_like(params[k]) for k in pro}
    nm.train_bind(params, grads)
    state, dis, tape = nm.train_prologue_fwd(vis.cuda(), emb.cuda(), gumbel.cuda())
    sf = lambda f, shape: native.state_field(state, B, T, f, shape)      # noqa: E731
    assert pc.maxdiff(sf(native.ST_K, (B, T, 512)), st["k"].permute(0, 2, 1)) < 2e-4
    assert pc.maxdiff(sf(native.ST_V, (B, T, 512)), st["v"]) < 2e-4
    assert pc.maxdiff(sf(native.ST_CKEY, (B, m, 256)), st["key"].permute(0, 2, 1)) < 2e-4
    assert pc.maxdiff(sf(native.ST_CVAL, (B, m, 256)), st["value"]) < 2e-4
    assert pc.maxdiff(sf(native.ST_ECELL, (B, 512)), st["encoder_cell"]) < 2e-4
    assert pc.maxdiff(dis, st["content_dis"]) < 1e-6
    g = {"dk": cot["k"].permute(0, 2, 1).contiguous().cuda(), "dv": cot["v"].cuda(), "dckey": cot["key"].permute(0, 2, 1).contiguous().cuda(),
         "dcval": cot["value"].cuda(), "dh_init": cot["hidden"].cuda(), "de_c": cot["encoder_cell"].cuda()}
    dvis = nm.train_prologue_bwd(vis.cuda(), emb.cuda(), state, tape, g, dcontent_dis=cot["content_dis"].cuda())

    def close(name, got, ref, rel=2e-3):
        ref = ref.reshape(got.shape)
        scale = max(1e-6, ref.abs().max().item())
        err = pc.maxdiff(got, ref) / scale
        assert err < rel, f"{name}: relative error {err:.2e} (scale {scale:.2e})"

    close("dvis", dvis, vis64.grad)
    bad = []
    for k in pro:
        assert sd64[k].grad is not None, k
        try:
            close(k, grads[k], sd64[k].grad)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, "\n".join(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,HW", [(2, 9, 96), (1, 29, 96), (1, 5, 88)])
def test_encoder_backward_matches_autograd(synth_sd, B, T, HW):
    """Visual encoder forward-with-tape and backward (front-end MaxPool/PReLU/BN + Conv3d weight gradient, the 16 ShuffleNet units,
    conv_last, AvgPool, L2-normalise) against autograd through the oracle's encoder: every encoder parameter.  The comparison runs
    the oracle in fp32 AND fp64: ReLU / MaxPool decisions on pre-activations within rounding of zero (or of each other) differ between an fp32
    and an fp64 forward and move single gradient entries by ~1e-2 (tools/dbg_enc_bwd.py: HIP vs fp32 oracle 1e-6, fp32 vs fp64 oracle
    1e-2 on the same entries); the fp64 oracle bounds the result at that level."""
    import parity_common as pc
    from lip2speech_amd import synth
    from oracle import l2s_oracle as orc
    video = synth.synth_video(B, T, HW, HW, tag=f"enc-train{T}") if HW != 96 else synth.synth_video(B, T, tag=f"enc-train{T}")
    torch.manual_seed(T)
    cot = torch.randn(B, T, 768, dtype=torch.float64)
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked"))      # noqa: E731
    enc = [k for k in synth_sd if k.startswith("encoder.")]
    par = [k for k in enc if synth_sd[k].is_floating_point() and not is_buf(k)]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = {}
    for dt in (torch.float64, torch.float32):
        sdx = {k: (synth_sd[k].detach().clone().to(dt).requires_grad_(k in par) if synth_sd[k].is_floating_point() else synth_sd[k]) for k in enc}
        feat_o = orc.encoder_forward(sdx, video.to(dt))
        (feat_o * cot.to(dt)).sum().backward()
        ref[dt] = {k: sdx[k].grad.double() for k in par}
        if dt == torch.float64:
            feat64 = feat_o.detach()
    nm = pc.native_model(synth_sd)
    params = {k: synth_sd[k].cuda() for k in par}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    nm.train_bind(params, grads)
    _, feat, tape = nm.train_encoder_fwd(video.cuda())
    assert pc.maxdiff(feat, feat64) < 2e-5
    dvis = torch.zeros(B, T, 1024, device="cuda")
    dvis[:, :, :768] = cot.float().cuda()
    nm.train_encoder_bwd(video.cuda(), dvis, tape)
    bad = []
    for k in par:
        r32, r64 = ref[torch.float32][k].reshape(grads[k].shape), ref[torch.float64][k].reshape(grads[k].shape)
        scale = max(1e-9, r64.abs().max().item())
        e32, e64 = pc.maxdiff(grads[k], r32) / scale, pc.maxdiff(grads[k], r64) / scale
        if not (min(e32, e64) < 2e-4 and max(e32, e64) < 5e-2):     # a flip separates the HIP forward from one oracle precision or the other
            bad.append(f"{k}: relative error vs fp32 oracle {e32:.2e}, vs fp64 oracle {e64:.2e} (scale {scale:.2e})")
    assert not bad, "\n".join(bad)


@pytest.mark.gpu
def test_dropout_sites_match_oracle(synth_sd):
    """Train-mode dropout as explicit multiplier inputs at the five sites of the reference (features, prenet, attention logits, LSTM
    inter-layer, post-net x5): the HIP decoder step with masks against autograd through the oracle with the same masks - outputs,
    loss terms, every decoder parameter gradient and the gradient wrt the visual features."""
    import parity_common as pc
    from lip2speech_amd import synth
    from lip2speech_amd.training import decoder_forward_backward, draw_dropout
    from oracle import l2s_oracle as orc
    B, T, S = 2, 29, 20
    gen = torch.Generator().manual_seed(77)
    drop = draw_dropout(B, T, S, "cpu", generator=gen)
    feat = torch.nn.functional.normalize(torch.randn(B, T, 768, generator=gen), dim=-1)
    emb = synth.synth_speaker_embedding(B, tag="drop")
    gum = synth.synth_gumbel(B * 4, tag="drop")
    mels = synth.synth_mels(B, S, tag="drop")
    gate = torch.zeros(B, S)
    gate[:, -1] = 1.0
    mask = torch.zeros(S, dtype=torch.bool)
    mask[[2, 3, 11]] = True
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked", "pos_table"))      # noqa: E731
    dec = [k for k in synth_sd if k.startswith("decoder.") and synth_sd[k].is_floating_point()]
    par = [k for k in dec if not is_buf(k)]
    sd64 = {k: synth_sd[k].detach().clone().double().requires_grad_(k in par) for k in dec}
    vis64 = orc.build_visual(feat, emb).double().requires_grad_(True)
    d64 = {k: ([m.double() for m in v] if isinstance(v, list) else v.double()) for k, v in drop.items()}
    vis_in = torch.cat([vis64[:, :, :768] * d64["feat"], vis64[:, :, 768:]], dim=2)
    st = orc.decoder_prologue(sd64, vis_in, emb.double(), gum.double())
    teacher = torch.cat([sd64["decoder.BOS"].view(1, 1, -1).expand(B, -1, -1), mels.double().permute(0, 2, 1)], dim=1)
    mel_o, stop_o, logit_o = orc.decode_loop(sd64, st, S, teacher=teacher, teacher_mask=mask, return_logits=True, drop=d64)
    mel_cf = mel_o.permute(0, 2, 1)
    post_o = orc.postnet(sd64, mel_cf, drop=d64["post"]) + mel_cf
    terms = orc.loss_terms([mel_cf, post_o, stop_o.unsqueeze(2), None, logit_o, st["content_dis"]], mels.double(), gate.double())
    terms[-1].backward()

    nm = pc.native_model(synth_sd)
    params = {k: synth_sd[k].cuda() for k in par}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    nm.train_bind(params, grads)
    cdrop = {k: ([m.cuda() for m in v] if isinstance(v, list) else v.cuda()) for k, v in drop.items()}
    out = decoder_forward_backward(nm, orc.build_visual(feat, emb).cuda(), emb.cuda(), gum.cuda(), mels.cuda(), gate.cuda(), teacher_mask=mask,
                                   bos=params["decoder.BOS"], drop=cdrop)
    assert pc.maxdiff(out["mel"], mel_cf) < 2e-4 and pc.maxdiff(out["mel_post"], post_o) < 5e-4
    assert pc.maxdiff(out["attn_logits"], logit_o) / logit_o.abs().max().item() < 1e-5
    want = torch.stack([t.detach() for t in terms])
    assert (out["loss"].cpu().double() - want).abs().max() < 2e-5 * want.abs().max()

    def rel(got, ref):
        ref = ref.reshape(got.shape)
        return pc.maxdiff(got, ref) / max(1e-6, ref.abs().max().item())
    assert rel(out["dvis"], vis64.grad) < 3e-3
    bad = [f"{k}: {rel(grads[k], sd64[k].grad):.2e}" for k in par if not k.startswith(("decoder.K.", "decoder.temperature", "decoder.Q."))
           and rel(grads[k], sd64[k].grad) > 3e-3]
    assert not bad, "\n".join(bad)


@pytest.mark.gpu
def test_encoder_batchnorm_train_mode(synth_sd):
    """nn.Module.train() semantics of the 56 BatchNorm layers of the encoder: forward with batch statistics (stats pass -> finalize -> fused
    kernel) and the running-statistics update against the oracle; the backward (a) is the gradient of that forward - central difference
    of the HIP forward along the HIP gradient, the direction with the largest signal - and (b) agrees with autograd through the oracle at
    the level at which the oracle agrees with itself across precisions: with statistics over 18 frames every ReLU / MaxPool decision
    that rounding flips moves the batch statistics of everything downstream (fp32 vs fp64 oracle: ~0.5 % L2 per tensor)."""
    import parity_common as pc
    from lip2speech_amd import native, synth
    from oracle import l2s_oracle as orc
    B, T = 2, 9
    video = synth.synth_video(B, T, tag="enc-bn-train")
    torch.manual_seed(11)
    cot = torch.randn(B, T, 768, dtype=torch.float64)
    enc = [k for k in synth_sd if k.startswith("encoder.")]
    is_stat = lambda k: k.endswith(("running_mean", "running_var"))      # noqa: E731
    par = [k for k in enc if synth_sd[k].is_floating_point() and not is_stat(k)]
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref, upd = {}, None
    for dt in (torch.float64, torch.float32):
        sdx = {k: (synth_sd[k].detach().clone().to(dt).requires_grad_(k in par) if synth_sd[k].is_floating_point() else synth_sd[k]) for k in enc}
        with orc.batch_statistics() as bs:
            feat_o = orc.encoder_forward(sdx, video.to(dt))
        (feat_o * cot.to(dt)).sum().backward()
        ref[dt] = {k: sdx[k].grad.double() for k in par}
        if dt == torch.float64:
            feat64, upd = feat_o.detach(), bs.updates
    native.set_option("refresh_map", 1)
    try:
        nm = native.NativeModel()
        nm.load({k: synth_sd[k].cuda() for k in enc}, enc)
    finally:
        native.set_option("refresh_map", 0)
    params = {k: synth_sd[k].clone().cuda() for k in enc if synth_sd[k].is_floating_point()}      # incl. the running statistics (updated in place)
    grads = {k: torch.zeros_like(params[k]) for k in par}
    nm.train_bind(params, grads)
    nm.train_set_bn(True, 0.1)
    vid, cotc = video.cuda(), cot.float().cuda()
    _, feat, tape = nm.train_encoder_fwd(vid)
    dvis = torch.zeros(B, T, 1024, device="cuda")
    dvis[:, :, :768] = cotc
    nm.train_encoder_bwd(vid, dvis, tape)
    assert pc.maxdiff(feat, feat64) < 5e-5
    assert len(upd) == 56
    for prefix, (rm, rv) in upd.items():
        assert pc.maxdiff(params[prefix + ".running_mean"], rm) < 1e-5 * max(1.0, rm.abs().max().item()), prefix
        assert pc.maxdiff(params[prefix + ".running_var"], rv) < 1e-5 * max(1.0, rv.abs().max().item()), prefix
    # (b) against the oracle
    worst = 0.0
    for k in par:
        r64, r32 = ref[torch.float64][k].reshape(grads[k].shape), ref[torch.float32][k].reshape(grads[k].shape)
        if r64.norm() < 1e-6 * max(1.0, ref[torch.float64]["encoder.trunk.1.0.weight"].norm().item()):
            continue                                     # exactly-zero gradients (a BN bias in front of another batch-stat BN): rounding residue
        err = min((grads[k].cpu().double() - r).norm().item() for r in (r64, r32)) / r64.norm().item()
        worst = max(worst, err)
        assert err < 4e-2, f"{k}: L2-relative error {err:.2e}"
    # the last layer sees no upstream flips: tight
    for k in ("encoder.trunk.1.0.weight", "encoder.trunk.1.1.weight", "encoder.trunk.1.1.bias"):
        r32 = ref[torch.float32][k].reshape(grads[k].shape)
        assert pc.maxdiff(grads[k], r32) < 1e-3 * r32.abs().max().item(), k
    # (a) central difference of the HIP forward along the HIP gradient
    g = {k: grads[k].clone() for k in par}
    base = {k: params[k].clone() for k in par}
    d = {k: g[k] * (base[k].pow(2).mean().sqrt() / (g[k].pow(2).mean().sqrt() + 1e-20)) for k in par}
    ana = sum(float((g[k].double() * d[k].double()).sum()) for k in par)
    eps, vals = 3e-6, []
    for sgn in (+1, -1):
        for k in par:
            params[k].copy_(base[k] + sgn * eps * d[k])
        nm.train_refresh_weights()
        _, f2, _ = nm.train_encoder_fwd(vid)
        vals.append(float((f2.double() * cotc.double()).sum()))
    fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - ana) < 2e-2 * abs(ana), (fd, ana, worst)


@pytest.mark.gpu
def test_decoder_batchnorm_train_mode_matches_autograd(synth_sd):
    """nn.Module.train() semantics of the decoder's 17 BatchNorm1d layers (8 MultiHop branches, 4 Content.agg branches, 5 post-net
    layers; all followed by smooth activations): the HIP decoder step with batch statistics against autograd through the oracle inside
    `batch_statistics()` - outputs, loss, running-statistics updates, every decoder parameter gradient, d(visual features)."""
    import parity_common as pc
    from lip2speech_amd import synth
    from lip2speech_amd.training import decoder_forward_backward
    from oracle import l2s_oracle as orc
    B, T, S = 2, 29, 20
    gen = torch.Generator().manual_seed(78)
    feat = torch.nn.functional.normalize(torch.randn(B, T, 768, generator=gen), dim=-1)
    emb = synth.synth_speaker_embedding(B, tag="bn-dec")
    gum = synth.synth_gumbel(B * 4, tag="bn-dec")
    mels = synth.synth_mels(B, S, tag="bn-dec")
    gate = torch.zeros(B, S)
    gate[:, -1] = 1.0
    is_stat = lambda k: k.endswith(("running_mean", "running_var"))      # noqa: E731
    dec = [k for k in synth_sd if k.startswith("decoder.") and synth_sd[k].is_floating_point()]
    par = [k for k in dec if not is_stat(k) and not k.endswith("pos_table")]
    sd64 = {k: synth_sd[k].detach().clone().double().requires_grad_(k in par) for k in dec}
    vis64 = orc.build_visual(feat, emb).double().requires_grad_(True)
    with orc.batch_statistics() as bs:
        st = orc.decoder_prologue(sd64, vis64, emb.double(), gum.double())
        mel_o, stop_o, logit_o = orc.decode_loop(sd64, st, S, return_logits=True)
        mel_cf = mel_o.permute(0, 2, 1)
        post_o = orc.postnet(sd64, mel_cf) + mel_cf
    terms = orc.loss_terms([mel_cf, post_o, stop_o.unsqueeze(2), None, logit_o, st["content_dis"]], mels.double(), gate.double())
    terms[-1].backward()
    assert len(bs.updates) == 17

    nm = pc.native_model(synth_sd)
    params = {k: synth_sd[k].clone().cuda() for k in dec}
    grads = {k: torch.zeros_like(params[k]) for k in par}
    nm.train_bind(params, grads)
    nm.train_set_bn(True, 0.1)
    try:
        out = decoder_forward_backward(nm, orc.build_visual(feat, emb).cuda(), emb.cuda(), gum.cuda(), mels.cuda(), gate.cuda())
    finally:
        nm.train_set_bn(False)
    assert pc.maxdiff(out["mel"], mel_cf) < 2e-4 and pc.maxdiff(out["mel_post"], post_o) < 5e-4
    want = torch.stack([t.detach() for t in terms])
    assert (out["loss"].cpu().double() - want).abs().max() < 2e-5 * want.abs().max()
    for prefix, (rm, rv) in bs.updates.items():
        assert pc.maxdiff(params[prefix + ".running_mean"], rm) < 1e-5 * max(1.0, rm.abs().max().item()), prefix
        assert pc.maxdiff(params[prefix + ".running_var"], rv) < 1e-5 * max(1.0, rv.abs().max().item()), prefix

    def rel(got, ref):
        ref = ref.reshape(got.shape)
        return pc.maxdiff(got, ref) / max(1e-6, ref.abs().max().item())
    assert rel(out["dvis"], vis64.grad) < 3e-3
    bad = [f"{k}: {rel(grads[k], sd64[k].grad):.2e}" for k in par if not k.startswith(("decoder.K.", "decoder.temperature", "decoder.Q."))
           and rel(grads[k], sd64[k].grad) > 3e-3]
    assert not bad, "\n".join(bad)
    # under batch statistics the bias of a conv in front of a BatchNorm has an exactly zero gradient
    for k in par:
        if k.endswith((".0.bias", ".0.conv.bias")) and (".conv." in k or ".agg." in k or "postnet.convolutions" in k):
            assert float(grads[k].abs().max()) == 0.0, k


@pytest.mark.gpu
def test_training_step_is_deterministic(synth_sd):
    """Two train()-mode forward + backward passes from the same parameters, inputs and masks give the same bits: outputs, loss terms and every
    parameter gradient (all reductions are two-stage with a fixed order, no atomics) - the property tools/hash_train_step.py relies on."""
    import parity_common as pc
    from lip2speech_amd import synth, training
    B, T, S = 4, 29, 24
    nm = pc.fresh_native_model(synth_sd)
    bound = {k: v.clone().cuda() for k, v in synth_sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point()}
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "pos_table"))      # noqa: E731
    grads = {k: torch.zeros_like(v) for k, v in bound.items() if not is_buf(k)}
    nm.train_bind(bound, grads)
    nm.train_set_bn(True, 0.1)
    video = synth.synth_video(B, T, tag="det").cuda(); emb = synth.synth_speaker_embedding(B, tag="det").cuda()
    gum = synth.synth_gumbel(B * 4, tag="det").cuda(); mels = synth.synth_mels(B, S, tag="det").cuda()
    gate = torch.zeros(B, S, device="cuda"); gate[:, -1] = 1
    torch.manual_seed(11)
    drop = training.draw_dropout(B, T, S, "cuda")
    mask = torch.zeros(S, dtype=torch.bool); mask[1::3] = True
    runs = []
    for _ in range(2):
        out = training.model_forward_backward(nm, video, emb, gum, mels, gate, teacher_mask=mask, bos=bound["decoder.BOS"], drop=drop)
        runs.append(({k: out[k].detach().clone() for k in ("loss", "mel", "mel_post", "stop")}, {k: v.clone() for k, v in grads.items()}))
    (o0, g0), (o1, g1) = runs
    assert all(torch.equal(o0[k], o1[k]) for k in o0), "outputs differ between two identical passes"
    bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    assert not bad, f"gradients differ between two identical passes: {bad[:5]}"
    assert all(torch.isfinite(v).all() for v in g0.values()) and float(sum(v.abs().sum() for v in g0.values())) > 0
