"""Shared helpers of the GPU parity tests: run the HIP path stage by stage next to the oracle."""
import os

import numpy as np
import torch

from lip2speech_amd import native, synth
from oracle import l2s_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def top2(a):
    srt, idx = torch.sort(a, dim=-1, descending=True)
    return idx[..., 0].to(torch.int32), srt[..., 0] - srt[..., 1]


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


_models = {}


def native_model(sd=None):
    """One packed NativeModel for the synthetic checkpoint (cached per process)."""
    if "m" not in _models:
        sd = synth.synth_state_dict() if sd is None else sd
        nm = native.NativeModel()
        nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
        _models["m"] = nm
    return _models["m"]


SHIPPED_PERSIST = 4          # the library's own default of "persist_decode" (l2s_common.h Options::persist); tests/conftest.py pins 0 for the suite


def shipped_model(sd=None):
    """The cached NativeModel with the options the library SHIPS with (persist_decode = 4: one to four clips of <= 32 frames take the persistent
    decode loop and, at one or two clips, the persistent BiLSTM) - what a caller of l2s_inference gets who sets no option."""
    if "shipped" not in _models:
        _models["shipped"] = fresh_native_model(sd, persist_decode=SHIPPED_PERSIST)
    return _models["shipped"]


class process_default:
    """with process_default("persist_decode", 4): models created inside (get_network(...) too) start with that option value."""
    def __init__(self, name, value, restore=0):
        self.name, self.value, self.restore = name, value, restore

    def __enter__(self):
        native.set_option(self.name, self.value)

    def __exit__(self, *exc):
        native.set_option(self.name, self.restore)


def fresh_native_model(sd=None, diag=None, **options):
    """A NativeModel of its own (not the cached one) with run-time options set BEFORE the weights are packed - options are per model.
    diag: bind the model to the diagnostic build (libl2s_diag.so: the block-form A/B switches of include/l2s_diag.h exist there only);
    default: exactly when one of `options` is such a switch."""
    sd = synth.synth_state_dict() if sd is None else sd
    if diag is None:
        diag = any(k in native.DIAG_OPTIONS for k in options)
    nm = native.NativeModel(native.diag() if diag else None)
    for k, v in options.items():
        nm.set_option(k, v)
    nm.load({k: v.cuda() for k, v in sd.items()}, list(sd.keys()))
    return nm


def lrw2_inputs():
    g = golden("inference_lrw_b2.npz")
    video = synth.synth_video(2, 29, tag="video-lrw2")
    emb = synth.synth_speaker_embedding(2, tag="spk-lrw2")
    return g, video, emb


def unfrag(frag, B, K):
    """frag16 buffer (flat) -> (B,K) on the host."""
    Bp = (B + 15) // 16 * 16
    f = frag.detach().cpu().view(Bp // 16, K // 16, 4, 16, 4)       # [rt][c][g][i][e]
    return f.permute(0, 3, 1, 2, 4).reshape(Bp, K)[:B]
