"""``Lip2Speech`` / ``get_network`` - the drop-in boundary (reference: /root/reference/model/model.py:13-72).

``get_network(mode)`` returns an ``nn.Module`` with ``.encoder``, ``.decoder``, ``.vgg_face`` attributes,
``forward(...)`` returning the reference's list of 7 and ``inference(...)`` returning
``(mel, output_lengths[, attention])``, so train.py / evaluate.py / demo.py style callers work unchanged.
``inference`` with a supplied ``speaker_embedding`` runs as ONE native call (``l2s_inference``: encoder,
prologue, 300-step loop and post-net enqueued back to back on the current stream).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .. import native, statespec
from .modules import Decoder, FaceRecognizer, VideoExtractor
from .modules._tree import NativeBacked

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class _HipTrainStep(torch.autograd.Function):
    """`Lip2Speech.forward` as ONE autograd node (reference: model.py:20-41 + decoder.py:320-379 under loss.backward(), train.py:184):
    forward runs the HIP kernels with tapes; backward runs the HIP backward (post-net, BPTT through the loop, prologue, encoder) and
    deposits every encoder/decoder parameter gradient in the flat gradient buffer the parameters' `.grad` alias.  `anchor` is any
    parameter of the model - it only makes autograd call `backward`."""

    @staticmethod
    def forward(ctx, anchor, model, video, emb, gumbel, mels, mask, drop):
        nm = model.native_model()
        B, _, T, _, _ = video.shape
        S = mels.shape[2]
        drop = drop or {}
        ctx.bn_batch = bool(model.training)                 # nn.Module.train(): batch statistics + running-statistics update (in the bound buffers)
        nm.train_set_bn(ctx.bn_batch)
        pdrop = native.postnet_drop_pack(drop["post"]) if drop.get("post") is not None else None
        vis, _, etape = nm.train_encoder_fwd(video, emb)
        if drop.get("feat") is not None:
            vis[:, :, :768] *= drop["feat"]                 # F.dropout(self.encoder(video_frames), 0.1, self.training), model.py:26
        state, dis, ptape = nm.train_prologue_fwd(vis, emb, gumbel)
        teacher = None
        if mask is not None:
            bos = model.decoder.BOS.detach().to(torch.float32).reshape(1, 1, -1).expand(B, 1, -1)
            teacher = torch.cat([bos, mels.detach().to(torch.float32).permute(0, 2, 1)[:, :S - 1]], dim=1).contiguous()
        (mel, stop, logits), sctx = nm.train_steps_fwd(state, B, T, S, teacher, mask, drop=drop)
        mel_post, post_tape = nm.train_postnet_fwd(mel, pdrop)
        ctx.model, ctx.tapes = model, (video, emb, vis, etape, state, ptape, sctx, mel, post_tape, pdrop, drop.get("feat"))
        ctx.mark_non_differentiable(logits)
        if ctx.bn_batch:
            model._count_batch()
        return mel.permute(0, 2, 1).contiguous(), mel_post, stop.unsqueeze(2), logits, dis

    @staticmethod
    def backward(ctx, dmel_cf, dmel_post, dstop, _dlogits, ddis):
        model = ctx.model
        nm = model.native_model()
        nm.train_set_bn(ctx.bn_batch)
        video, emb, vis, etape, state, ptape, sctx, mel, post_tape, pdrop, fdrop = ctx.tapes
        B, S = mel.shape[0], mel.shape[1]
        flat = model._flat
        # a fused zero_grad() just zeroed the buffer and nothing has written it through torch since (FlatBuffer.cleared_version): nothing to
        # accumulate.  A manual p.grad.add_(), an all-reduce or another graph's backward in between bumps the version -> the accumulate path
        live = model._grads_live() and not flat.is_cleared()
        flat.mark_written()
        prev = flat.grad.clone() if live else None                    # gradient accumulation across backward() calls
        n_dec = model._n_decoder_elems()
        z = lambda g, ref: torch.zeros_like(ref) if g is None else g  # noqa: E731
        wbuf = nm.train_pack_weights(video.device)
        dmel = nm.train_postnet_bwd(mel, z(dmel_post, mel.permute(0, 2, 1)), post_tape, pdrop)
        if dmel_cf is not None:
            dmel += dmel_cf.permute(0, 2, 1)
        sg = nm.train_steps_bwd(sctx, dmel, z(dstop, mel[:, :, :1]).reshape(B, S), wbuf=wbuf)
        dvis = nm.train_prologue_bwd(vis, emb, state, ptape, sg, dcontent_dis=ddis, wbuf=wbuf)
        if fdrop is not None:
            dvis[:, :, :768] *= fdrop
        # every decoder gradient is final here (the flat buffer holds the decoder group first): a data-parallel caller's hook starts the
        # all-reduce of those buckets now, under the encoder backward (train.py:184-193's hook point; bench.py --mode train does the same)
        if prev is not None:
            flat.grad[:n_dec] += prev[:n_dec]
        hook = model.__dict__.get("_on_decoder_grads")
        if hook is not None:
            hook()
        nm.train_encoder_bwd(video, dvis, etape)
        if prev is not None:
            flat.grad[n_dec:] += prev[n_dec:]
        model._attach_grads()
        return (None,) * 8


class Lip2Speech(NativeBacked):
    _key_prefix = ""

    def __init__(self):
        super().__init__()
        self._init_native()
        self.vgg_face = FaceRecognizer()
        self.encoder = VideoExtractor()
        self.decoder = Decoder()
        # one packed weight blob for the whole path; the sub-modules borrow it while they live inside this model
        self.encoder.__dict__["_native_parent"] = self
        self.decoder.__dict__["_native_parent"] = self

    def _collect_tensors(self):
        sd = self.state_dict(keep_vars=True)
        return {k: v for k, v in sd.items() if k.startswith(("encoder.", "decoder."))}

    # ------------------------------------------------------------------ training state
    def trainable_groups(self):
        """The optimizer's parameter groups as train.py:102-104 builds them: decoder, then encoder."""
        return [list(self.decoder.parameters()), list(self.encoder.parameters())]

    def _train_state(self):
        """Lazy: re-home the decoder/encoder parameters into one flat fp32 buffer (+ a flat gradient buffer) and bind both to the library."""
        if self.__dict__.get("_flat") is None:
            from ..training import FlatBuffer
            flat = FlatBuffer(self.trainable_groups())
            self.__dict__["_flat"] = flat
            names = {id(p): "decoder." + n for n, p in self.decoder.named_parameters()}
            names.update({id(p): "encoder." + n for n, p in self.encoder.named_parameters()})
            self.__dict__["_flat_names"] = [names[id(p)] for p in flat.params]
            grads = {}
            for p, key in zip(flat.params, self._flat_names):
                off = flat.offsets[id(p)]
                grads[key] = flat.grad[off:off + p.numel()].view_as(p)
                p.grad = None
            self.__dict__["_grad_views"] = grads
            # the (re)load below also builds the device-side refresh map (a per-model option: other models keep the cheap load)
            nm = self.__dict__.get("_native") or native.NativeModel()
            nm.set_option("refresh_map", 1)
            self.__dict__["_native"] = nm
            self.__dict__["_native_sig"] = None
            nm = self.native_model()
            bound = {k: p.data for k, p in zip(self._flat_names, flat.params)}
            bound.update({k: v for k, v in self._tensors().items() if k not in bound and v.is_floating_point()})   # buffers: BN statistics, pos_table
            nm.train_bind(bound, grads)
            self.__dict__["_refresh_on_device"] = True
        return self._flat

    def _n_decoder_elems(self) -> int:
        """Elements of the decoder group = the leading range of the flat parameter / gradient buffers."""
        if self.__dict__.get("_n_dec") is None:
            self.__dict__["_n_dec"] = sum(p.numel() for p in self.decoder.parameters())
        return self.__dict__["_n_dec"]

    def _count_batch(self):
        """num_batches_tracked += 1 on every BatchNorm (what nn.BatchNorm does in train mode; the running statistics themselves are updated
        by the kernels)."""
        if self.__dict__.get("_nbt") is None:
            self.__dict__["_nbt"] = [b for n, b in self.named_buffers() if n.endswith("num_batches_tracked") and n.startswith(("encoder.", "decoder."))]
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)

    def _grads_live(self) -> bool:
        return any(p.grad is not None for p in self._flat.params)

    def _attach_grads(self):
        for p, key in zip(self._flat.params, self._flat_names):
            p.grad = self._grad_views[key]

    def _speaker(self, face_frames, speaker_embedding):
        if speaker_embedding is not None:
            return speaker_embedding
        return self.vgg_face.inference(face_frames[:, 0, :, :, :])

    def forward(self, video_frames, face_frames, audio_frames, melspecs, video_lengths, audio_lengths, melspec_lengths,
                tf_ratio, speaker_embedding=None, gumbel_noise=None, dropout_masks=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.decoder.parameters()):
            return self._forward_train(video_frames, face_frames, melspecs, video_lengths, tf_ratio, speaker_embedding, gumbel_noise, dropout_masks)
        if self.training:
            # a no_grad call on a train()-mode module: the feature dropout of model.py:26 is live - the staged route through the sub-modules
            video_features = F.dropout(self.encoder(video_frames), 0.1, True)
            emb = self._speaker(face_frames, speaker_embedding)
            vis = native.build_visual(video_features, emb)
            face = emb.unsqueeze(1).expand(-1, vis.shape[1], -1)
            outputs = self.decoder(vis, face, melspecs, video_lengths, melspec_lengths, tf_ratio, gumbel_noise=gumbel_noise)
            return outputs + [video_lengths]
        job = self._forward_job(video_frames, face_frames, melspecs, video_lengths, tf_ratio, speaker_embedding, gumbel_noise)
        out = self.native_model().forward_eval(job["video"], job["emb"], job["gumbel"], job["S"], teacher=job.get("teacher"), teacher_mask=job.get("mask"))
        return job["finish"](out)

    def _forward_job(self, video_frames, face_frames, melspecs, video_lengths, tf_ratio, speaker_embedding=None, gumbel_noise=None):
        """One eval-mode `forward` call as a job of the native path (`l2s_forward_eval`; model.py:23-40 + decoder.py:320-379): the speaker
        embedding, the Gumbel noise (drawn on the device unless supplied) and the scheduled-sampling mask (one host `torch.rand(1)` per
        step) are fixed here, in the reference's order; `finish` shapes the native outputs into the reference's list of 7."""
        dev = self.decoder.BOS.device
        with torch.no_grad():
            video = video_frames.to(dev, non_blocking=True)
            emb = self._speaker(face_frames.to(dev, non_blocking=True) if face_frames is not None else None,
                                speaker_embedding.to(dev, non_blocking=True) if speaker_embedding is not None else None)
            B, _, T, _, _ = video.shape
            S = melspecs.shape[2]
            if gumbel_noise is None:
                gumbel_noise = Decoder.draw_gumbel(B * native.min_T(T), dev)
            mask = Decoder.sampling_mask(S, tf_ratio)
            job = {"entry": "forward", "video": video, "emb": emb, "gumbel": gumbel_noise.to(dev, non_blocking=True), "S": S, "mask": mask}
            if mask is not None:
                job["teacher"] = self.decoder.teacher_frames(melspecs.to(dev, non_blocking=True))
        job["finish"] = lambda o: [o[0], o[1], o[2].unsqueeze(2), emb, o[3], o[4], video_lengths]
        return job

    def _inference_job(self, video_frames, face_frames, speaker_embedding=None, return_attention_map=False, gumbel_noise=None):
        dev = self.decoder.BOS.device
        with torch.no_grad():
            video = video_frames.to(dev, non_blocking=True)
            emb = self._speaker(face_frames.to(dev, non_blocking=True) if face_frames is not None else None,
                                speaker_embedding.to(dev, non_blocking=True) if speaker_embedding is not None else None)
            B, _, T, _, _ = video.shape
            if gumbel_noise is None:
                gumbel_noise = Decoder.draw_gumbel(B * native.min_T(T), dev)
        return {"entry": "inference", "video": video, "emb": emb, "gumbel": gumbel_noise.to(dev, non_blocking=True),
                "S": self.decoder.hparams.max_decoder_steps, "want_attn": bool(return_attention_map),
                "finish": (lambda o: (o[0], o[1], o[2])) if return_attention_map else (lambda o: (o[0], o[1]))}

    # ------------------------------------------------------------------ loader-driven callers: G batches per launch chain, chains in flight
    def pool(self, group: int = 8, n_inflight: int = 3):
        """The `parallel.InflightPool` of this model's weight blob (cached per shape of concurrency)."""
        from ..parallel import InflightPool
        nm = self.native_model()
        pools = self.__dict__.setdefault("_pools", {})
        key = (group, n_inflight)
        if key not in pools or pools[key].model is not nm:      # a re-packed / replaced NativeModel: the old pool (and its blob) is dropped
            pools[key] = InflightPool(model=nm, n_inflight=n_inflight, group=group, device=self.decoder.BOS.device)
        for k in [k for k, v in pools.items() if v.model is not nm]:
            del pools[k]
        return pools[key]

    @staticmethod
    def _call(item):
        """An item of `*_many`: the positional arguments of the single-batch method, optionally followed by a dict of keyword arguments."""
        if isinstance(item, dict):
            return (), item
        item = tuple(item)
        if item and isinstance(item[-1], dict):
            return item[:-1], item[-1]
        return item, {}

    def inference_many(self, calls, group: int = 8, n_inflight: int = 3):
        """`inference` over a stream of batches - the loop of demo.py:60-90 - with `group` batches advanced per launch chain
        (`l2s_inference_multi`) and `n_inflight` chains on the GPU at once.  `calls` is any iterable (e.g. a generator over a DataLoader) of
        argument tuples `(video_frames, face_frames[, speaker_embedding[, return_attention_map[, gumbel_noise]]])`, optionally ending in a
        dict of keyword arguments; inputs may live on the host (they are copied on the pool's copy stream, one group ahead).  Yields, in
        order, exactly what `inference(*call)` returns for each - bit-identical when the Gumbel noise is supplied."""
        self.native_model()
        prep = lambda item: (lambda a, k: self._inference_job(*a, **k))(*self._call(item))      # noqa: E731
        return self.pool(group, n_inflight).imap(calls, prep)

    def forward_many(self, calls, group: int = 8, n_inflight: int = 3):
        """Eval-mode `forward` over a stream of batches - the loop of evaluate.py:22-51 (`net(..., tf_ratio=1)` per DataLoader batch) - on the
        grouped path (`l2s_forward_eval_multi`).  `calls`: iterable of `forward`'s argument tuples `(video_frames, face_frames, audio_frames,
        melspecs, video_lengths, audio_lengths, melspec_lengths, tf_ratio)`, optionally ending in a dict (`speaker_embedding=`,
        `gumbel_noise=`).  Yields the reference's list of 7 per call, in order, bit-identical to `forward(*call)` under `torch.no_grad()`.
        Batches group when they share shape, S and scheduled-sampling mask (always at tf_ratio = 1: no step is teacher-forced)."""
        if self.training:
            raise RuntimeError("forward_many is the evaluate path: call .eval() first (train() mode trains through forward / backward)")
        self.native_model()

        def prep(item):
            a, k = self._call(item)
            video, face, _audio, mels, vlen, _alen, _mlen, tf = a
            return self._forward_job(video, face, mels, vlen, tf, **k)
        return self.pool(group, n_inflight).imap(calls, prep)

    def _forward_train(self, video_frames, face_frames, melspecs, video_lengths, tf_ratio, speaker_embedding, gumbel_noise, dropout_masks=None):
        """The differentiable route (train.py:167-184): same outputs as `forward`, attached to autograd through `_HipTrainStep`.
        In `train()` mode the five dropout sites of the reference are active (multipliers drawn on the device by `training.draw_dropout`
        and handed to the kernels as inputs, or supplied through `dropout_masks=`); in `eval()` mode they are off (the configuration the
        reference gradient goldens pin).  BatchNorm follows the module mode too: batch statistics with running-statistics updates in
        `train()`, running statistics in `eval()` (per process: no cross-rank synchronisation, like the single-device reference)."""
        self._train_state()
        with torch.no_grad():
            emb = self._speaker(face_frames, speaker_embedding).to(torch.float32).contiguous()
        B, _, T, _, _ = video_frames.shape
        S = melspecs.shape[2]
        if gumbel_noise is None:
            gumbel_noise = Decoder.draw_gumbel(B * native.min_T(T), video_frames.device)
        mask, consumed = [], 0
        for _ in range(S):                                   # scheduled sampling, one torch.rand(1) per step (decoder.py:355-357)
            take = bool(torch.rand(1) > tf_ratio) and consumed < int(tf_ratio * S)
            consumed += int(take)
            mask.append(1 if take else 0)
        drop = dropout_masks
        if drop is None and self.training:
            from ..training import draw_dropout
            drop = draw_dropout(B, T, S, video_frames.device)
        mel, mel_post, stop, attn, dis = _HipTrainStep.apply(self.decoder.BOS, self, video_frames.detach().to(torch.float32).contiguous(), emb,
                                                             gumbel_noise.detach().to(torch.float32).contiguous(), melspecs, mask if any(mask) else None, drop)
        return [mel, mel_post, stop, emb, attn, dis, video_lengths]

    def inference(self, video_frames, face_frames, speaker_embedding=None, return_attention_map=False, gumbel_noise=None):
        with torch.no_grad():
            emb = self._speaker(face_frames, speaker_embedding)
            B, _, T, _, _ = video_frames.shape
            if gumbel_noise is None:
                gumbel_noise = Decoder.draw_gumbel(B * native.min_T(T), video_frames.device)
            mel, lengths, attn = self.native_model().inference(
                video_frames, emb, gumbel_noise, S=self.decoder.hparams.max_decoder_steps, want_attn=return_attention_map)
        if return_attention_map:
            return mel, lengths, attn
        return mel, lengths


def inference_pool(net: "Lip2Speech", n_inflight: int = 2, group: int = 4):
    """Throughput serving for a loaded model: a `parallel.InflightPool` over the model's own packed weight blob.  `pool.map(
    [(video, speaker_embedding, gumbel_noise), ...])` advances `group` independent batches per launch chain (`l2s_inference_multi`) with
    `n_inflight` chains on the GPU at once (2.4x the one-at-a-time throughput on one MI355X at group 8) and returns
    `(mel, output_lengths, attention)` per batch, each identical to `net.inference` on that batch."""
    from ..parallel import InflightPool
    return InflightPool(model=net.native_model(), n_inflight=n_inflight, group=group)


def get_network(mode):
    assert mode in ("train", "test")
    model = Lip2Speech()
    return model.train() if mode == "train" else model.eval()
