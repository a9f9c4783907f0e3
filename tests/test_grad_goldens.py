"""Gradient goldens from the imported reference (tests/golden/make_grad_goldens.py): eval-mode
`forward(tf_ratio)` + `Loss` + `backward()`.  CPU: the oracle's autograd reproduces them (pins the oracle as the checker of the
backward kernels).  GPU: the HIP training path reproduces them."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_common as pc          # noqa: E402
from lip2speech_amd import synth    # noqa: E402
from oracle import l2s_oracle as orc  # noqa: E402

CASES = ["lrw_b2_s77_tf1", "lrw_b2_s77_tf05"]
B, T, S = 2, 29, 77


def projection_vector(key: str, n: int) -> np.ndarray:
    return np.where(synth.uniform01("gradproj:" + key, n) < 0.5, -1.0, 1.0)


def gate_targets():
    g = torch.zeros(B, S)
    g[:, S - 1] = 1.0
    return g


def load(case):
    z = np.load(os.path.join(pc.GOLDEN, f"grads_{case}.npz"))
    return z, {k: i for i, k in enumerate(z["keys"].tolist())}


# Gradients that reach a parameter only through the attention soft-max over T.  The temperature MULTIPLIES (logits ~ +-6000), so that
# soft-max is one-hot to fp32 precision at almost every step and the reference's own fp32 gradient there is ~1e-7 of rounding residue
# (total gradient norm: 1.1e3).  These tensors are compared with an absolute floor instead of relatively.
SATURATED = ("decoder.K.", "decoder.temperature", "decoder.Q.")
SATURATED_FLOOR = 3e-6


def check_grads(z, index, grads, keys, rel_norm, rel_proj, rel_full):
    """grads: key -> tensor.  norm within rel_norm; +-1 projection within rel_proj of the norm; tensors <= 2048 elements in full."""
    bad = []
    for k in keys:
        floor = SATURATED_FLOOR if k.startswith(SATURATED) else 0.0
        g = grads[k].detach().double().cpu().numpy().ravel()
        i = index[k]
        n_ref, p_ref = float(z["grad_norm"][i]), float(z["grad_proj"][i])
        n = float(np.sqrt((g * g).sum()))
        p = float((g * projection_vector(k, g.size)).sum())
        scale = max(n_ref, 1e-12) + floor / rel_norm
        if abs(n - n_ref) > rel_norm * scale:
            bad.append(f"{k}: norm {n:.6e} vs {n_ref:.6e}")
        # |proj error| <= ||dg||_1-ish; compare against the norm times sqrt(size) damped: a +-1 projection of an error vector e has
        # magnitude ~ ||e||_2, so bound it by rel_proj * ||g||_2
        if abs(p - p_ref) > rel_proj * scale * max(1.0, np.sqrt(np.log(g.size + 1.0))):
            bad.append(f"{k}: projection {p:.6e} vs {p_ref:.6e} (norm {n_ref:.3e})")
        if "grad:" + k in z.files:
            full = z["grad:" + k].astype(np.float64).ravel()
            err = np.abs(g - full).max()
            if err > rel_full * max(np.abs(full).max(), 1e-12) + floor:
                bad.append(f"{k}: full-gradient max error {err:.3e} (scale {np.abs(full).max():.3e})")
    assert not bad, "\n".join(bad[:40]) + f"\n... {len(bad)} mismatches"


@pytest.mark.parametrize("case", CASES)
def test_oracle_autograd_matches_reference_gradients(case):
    """fp64 autograd through the oracle vs the reference's own fp32 backward: every encoder and decoder parameter."""
    z, index = load(case)
    sd = synth.synth_state_dict()
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked", "pos_table"))      # noqa: E731
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buf(k) and not k.startswith(("speaker_encoder.", "vgg_face.")) else
                (v.double() if v.is_floating_point() else v)) for k, v in sd.items()}
    video = synth.synth_video(B, T, tag="video-lrw2").double().requires_grad_(True)
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2").double()
    gum = synth.synth_gumbel(B * 4, tag="gumbel-lrw2").double()
    mels = synth.synth_mels(B, S, tag="mel-lrw2").double()
    mask = torch.from_numpy(z["teacher_mask"])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    outs = orc.forward_eval(sd64, video, emb, mels, gum, teacher_mask=mask if bool(mask.any()) else None)
    terms = orc.loss_terms(outs, mels, gate_targets().double())
    terms[-1].backward()
    got = np.asarray([t.item() for t in terms])
    assert np.abs(got - z["loss_terms"]).max() < 1e-5 * np.abs(z["loss_terms"]).max()
    grads = {k: sd64[k].grad for k in index}
    assert all(g is not None for g in grads.values())
    check_grads(z, index, grads, list(index), rel_norm=2e-3, rel_proj=4e-3, rel_full=4e-3)
    dv = video.grad.numpy()
    assert abs(np.sqrt((dv * dv).sum()) - float(z["dvideo_norm"])) < 2e-3 * float(z["dvideo_norm"])
    assert np.abs(dv[0, :, 14] - z["dvideo_clip0_frame14"]).max() < 4e-3 * np.abs(z["dvideo_clip0_frame14"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_decoder_training_step_matches_reference_gradients(case):
    """The HIP training path of the decoder (prologue -> 77-step loop -> post-net -> 4-term loss -> backward through all of it)
    against the reference's own backward: loss terms, every decoder parameter gradient, and the gradient wrt the encoder features."""
    from lip2speech_amd import native
    from lip2speech_amd.training import decoder_forward_backward
    z, index = load(case)
    sd = synth.synth_state_dict()
    g = pc.golden("inference_lrw_b2.npz")
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2").cuda()
    gum = synth.synth_gumbel(B * 4, tag="gumbel-lrw2").cuda()
    mels = synth.synth_mels(B, S, tag="mel-lrw2").cuda()
    nm = pc.native_model(sd)
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked", "pos_table"))      # noqa: E731
    params = {k: v.cuda() for k, v in sd.items() if k.startswith("decoder.") and v.is_floating_point() and not is_buf(k)}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    nm.train_bind(params, grads)
    vis = native.build_visual(g["feat"].cuda(), emb)
    mask = torch.from_numpy(z["teacher_mask"])
    out = decoder_forward_backward(nm, vis, emb, gum, mels, gate_targets().cuda(), teacher_mask=mask if bool(mask.any()) else None,
                                   bos=params["decoder.BOS"])
    assert pc.maxdiff(out["mel"], torch.from_numpy(z["mel"])) < 1e-3
    assert pc.maxdiff(out["stop"], torch.from_numpy(z["stop"])) < 1e-3
    got = out["loss"].cpu().double().numpy()
    assert np.abs(got - z["loss_terms"]).max() < 2e-5 * np.abs(z["loss_terms"]).max(), (got, z["loss_terms"])
    dec_keys = [k for k in index if k.startswith("decoder.")]
    assert set(dec_keys) == set(grads), set(dec_keys) ^ set(grads)
    check_grads(z, index, grads, dec_keys, rel_norm=3e-3, rel_proj=6e-3, rel_full=6e-3)
    dfeat = out["dvis"][:, :, :768]
    ref = torch.from_numpy(z["dfeat"])
    assert pc.maxdiff(dfeat, ref) < 4e-3 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_full_training_step_matches_reference_gradients(case):
    """Whole model: video -> encoder -> decoder -> loss -> backward through the decoder AND the encoder, against the reference's own
    backward: all 308 encoder/decoder parameter gradients."""
    from lip2speech_amd.training import model_forward_backward
    z, index = load(case)
    sd = synth.synth_state_dict()
    video = synth.synth_video(B, T, tag="video-lrw2").cuda()
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2").cuda()
    gum = synth.synth_gumbel(B * 4, tag="gumbel-lrw2").cuda()
    mels = synth.synth_mels(B, S, tag="mel-lrw2").cuda()
    nm = pc.native_model(sd)
    params = {k: sd[k].cuda() for k in index}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    nm.train_bind(params, grads)
    mask = torch.from_numpy(z["teacher_mask"])
    out = model_forward_backward(nm, video, emb, gum, mels, gate_targets().cuda(), teacher_mask=mask if bool(mask.any()) else None,
                                 bos=params["decoder.BOS"])
    got = out["loss"].cpu().double().numpy()
    assert np.abs(got - z["loss_terms"]).max() < 2e-5 * np.abs(z["loss_terms"]).max(), (got, z["loss_terms"])
    check_grads(z, index, grads, list(index), rel_norm=3e-3, rel_proj=6e-3, rel_full=6e-3)
    total = float(np.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values())))
    ref_total = float(np.sqrt((z["grad_norm"] ** 2).sum()))
    assert abs(total - ref_total) < 1e-3 * ref_total           # what clip_grad_norm_ sees (train.py:191)


# ---------------------------------------------------------------------------------------------------------------------------------
# train() mode BatchNorm (batch statistics + running-statistics update) with the dropout sites off: golden `lrw_b2_s77_bntrain`.
# Encoder gradients: with statistics over 58 frames, every ReLU / MaxPool decision that rounding flips shifts the batch statistics of
# everything downstream, so two correct implementations agree to ~1 % per tensor (fp32 vs fp64 oracle: 0.5 %); they are compared at
# that level.  The decoder's BatchNorm layers are followed by smooth activations and are compared like the eval-mode goldens.
def check_bntrain(z, index, grads, loss_terms, buffers):
    assert np.abs(np.asarray(loss_terms) - z["loss_terms"]).max() < 1e-4 * np.abs(z["loss_terms"]).max(), (loss_terms, z["loss_terms"])
    for name in z.files:
        if name.startswith("buf:"):
            got = buffers[name[4:]].detach().double().cpu().numpy()
            assert np.abs(got - z[name]).max() < 2e-4 * max(1.0, np.abs(z[name]).max()), name      # downstream of the 77-step fp32 recurrence
    dec_keys = [k for k in index if k.startswith("decoder.")]
    # biases of convs in front of a batch-statistics BatchNorm: exactly zero gradient, the reference holds rounding residue there
    zero = [k for k in dec_keys if k.endswith((".0.bias", ".0.conv.bias")) and (".conv." in k or ".agg." in k or "postnet.convolutions" in k)]
    check_grads(z, index, grads, [k for k in dec_keys if k not in zero], rel_norm=4e-3, rel_proj=8e-3, rel_full=8e-3)
    total = float(np.sqrt((z["grad_norm"] ** 2).sum()))
    for k in zero:
        assert float(grads[k].detach().double().norm()) < 1e-6 * total and float(z["grad_norm"][index[k]]) < 1e-5 * total, k
    bad = []
    for k in index:
        if not k.startswith("encoder."):
            continue
        g = grads[k].detach().double().cpu().numpy().ravel()
        n_ref = float(z["grad_norm"][index[k]])
        if n_ref < 1e-7 * total:
            continue
        n = float(np.sqrt((g * g).sum()))
        if abs(n - n_ref) > 4e-2 * n_ref:
            bad.append(f"{k}: norm {n:.5e} vs {n_ref:.5e}")
        if "grad:" + k in z.files:
            full = z["grad:" + k].astype(np.float64).ravel()
            e = float(np.sqrt(((g - full) ** 2).sum())) / max(n_ref, 1e-30)
            if e > 5e-2:
                bad.append(f"{k}: L2-relative error {e:.2e}")
    assert not bad, "\n".join(bad)


def test_oracle_batch_statistics_match_reference_train_mode():
    """CPU: the oracle inside `batch_statistics()` against the reference's own train() forward + backward."""
    z, index = load("lrw_b2_s77_bntrain")
    sd = synth.synth_state_dict()
    is_buf = lambda k: k.endswith(("running_mean", "running_var", "num_batches_tracked", "pos_table"))      # noqa: E731
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not is_buf(k) and not k.startswith(("speaker_encoder.", "vgg_face.")) else
                (v.double() if v.is_floating_point() else v)) for k, v in sd.items()}
    video = synth.synth_video(B, T, tag="video-lrw2").double()
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2").double()
    gum = synth.synth_gumbel(B * 4, tag="gumbel-lrw2").double()
    mels = synth.synth_mels(B, S, tag="mel-lrw2").double()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with orc.batch_statistics() as bs:
        outs = orc.forward_eval(sd64, video, emb, mels, gum)
    terms = orc.loss_terms(outs, mels, gate_targets().double())
    terms[-1].backward()
    buffers = {}
    for prefix, (rm, rv) in bs.updates.items():
        buffers[prefix + ".running_mean"], buffers[prefix + ".running_var"] = rm, rv
    assert len(bs.updates) == 73
    check_bntrain(z, index, {k: sd64[k].grad for k in index}, [t.item() for t in terms], buffers)


@pytest.mark.gpu
def test_hip_train_mode_batchnorm_matches_reference():
    """GPU: the HIP training step with batch-statistics BatchNorm (all 73 layers) against the reference's own train() forward + backward."""
    from lip2speech_amd.training import model_forward_backward
    z, index = load("lrw_b2_s77_bntrain")
    sd = synth.synth_state_dict()
    video = synth.synth_video(B, T, tag="video-lrw2").cuda()
    emb = synth.synth_speaker_embedding(B, tag="spk-lrw2").cuda()
    gum = synth.synth_gumbel(B * 4, tag="gumbel-lrw2").cuda()
    mels = synth.synth_mels(B, S, tag="mel-lrw2").cuda()
    nm = pc.native_model(sd)
    bound = {k: v.clone().cuda() for k, v in sd.items() if k.startswith(("encoder.", "decoder.")) and v.is_floating_point()}
    grads = {k: torch.zeros_like(bound[k]) for k in index}
    nm.train_bind(bound, grads)
    nm.train_set_bn(True, 0.1)
    try:
        out = model_forward_backward(nm, video, emb, gum, mels, gate_targets().cuda())
    finally:
        nm.train_set_bn(False)
    check_bntrain(z, index, grads, out["loss"].cpu().double().numpy(), bound)
