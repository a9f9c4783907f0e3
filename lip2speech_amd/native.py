"""ctypes binding of libl2s_hip.so (the C-ABI declared in include/l2s.h).

PyTorch is used here for what it is good at on the host side of this path - device
memory (caching allocator), streams, ``torch.distributed`` - and nothing else: every
function below hands raw device pointers and the current HIP stream to the native
library.  There is no fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised.
"""
from __future__ import annotations

import collections
import ctypes
import os
import threading
from typing import Dict, Iterable, Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DIAG_LIB_PATH = os.path.join(_HERE, "csrc", "libl2s_diag.so")      # the diagnostic build (include/l2s_diag.h): tools/ and the operator tests only
# L2S_LIB: another build of the same ABI - tools/ point it at the diagnostic library ("diag" = DIAG_LIB_PATH), whose ABI is a superset
LIB_PATH = os.environ.get("L2S_LIB") or os.path.join(_HERE, "csrc", "libl2s_hip.so")
if LIB_PATH == "diag":
    LIB_PATH = DIAG_LIB_PATH

# every symbol include/l2s.h declares; tests check the built library exports all of them (and the product library nothing else)
ABI_SYMBOLS = (
    "l2s_abi_version", "l2s_last_error",
    "l2s_model_create", "l2s_model_set_tensor", "l2s_model_finalize", "l2s_model_destroy",
    "l2s_min_T", "l2s_workspace_bytes", "l2s_state_floats", "l2s_state_offset",
    "l2s_encoder_fwd", "l2s_normalise_pad_frames", "l2s_build_visual", "l2s_decoder_prologue", "l2s_decode_steps", "l2s_postnet",
    "l2s_output_lengths", "l2s_inference", "l2s_inference_multi", "l2s_workspace_bytes_multi", "l2s_forward_eval", "l2s_forward_eval_multi", "l2s_model_set_option", "l2s_persist_available", "l2s_persist_timeouts", "l2s_set_thread_chains", "l2s_speaker_workspace_bytes", "l2s_speaker_encoder_fwd",
    "l2s_inverse_mel_workspace_bytes", "l2s_inverse_mel", "l2s_griffin_lim_workspace_bytes", "l2s_griffin_lim", "l2s_estoi_workspace_bytes", "l2s_estoi",
    "l2s_set_option",
    "l2s_train_scratch_bytes", "l2s_loss", "l2s_grad_norm", "l2s_adamw_amsgrad_step",
    "l2s_train_steps_tape_floats", "l2s_train_steps_weights_floats", "l2s_train_steps_ws_bytes", "l2s_train_steps_pack_weights",
    "l2s_train_steps_fwd", "l2s_train_steps_bwd",
    "l2s_train_set_bn", "l2s_train_refresh_weights", "l2s_train_encoder_tape_floats", "l2s_train_encoder_ws_bytes", "l2s_train_encoder_fwd", "l2s_train_encoder_bwd",
    "l2s_train_prologue_tape_floats", "l2s_train_prologue_ws_bytes", "l2s_train_prologue_fwd", "l2s_train_prologue_bwd",
    "l2s_train_bind", "l2s_train_postnet_tape_floats", "l2s_train_postnet_ws_bytes", "l2s_train_postnet_fwd", "l2s_train_postnet_bwd",
    "l2s_profile_enable", "l2s_profile_reset", "l2s_profile_count", "l2s_profile_get",
)
# what include/l2s_diag.h adds: exported by libl2s_diag.so only
DIAG_SYMBOLS = (
    "l2s_op_gemm", "l2s_op_conv1d", "l2s_op_gemm_ex", "l2s_op_conv1d_ex", "l2s_op_conv1d_bwd", "l2s_op_frontend", "l2s_op_launch_chain", "l2s_op_launch_chain2", "l2s_op_skinny_timeline", "l2s_op_attn_timeline", "l2s_op_flat_timeline", "l2s_op_pdecode_timeline", "l2s_op_gemm_x3_timeline", "l2s_op_fused_unit_timeline", "l2s_op_lstm_cell_chain", "l2s_op_step_attn_chain", "l2s_op_stamp_log",
)

# run-time options only the diagnostic build accepts (block-form A/B switches of the same arithmetic, include/l2s_diag.h)
DIAG_OPTIONS = frozenset(("overlap_postnet", "fuse_trunk", "fuse_s2", "skinny_static", "skinny_sized", "skinny_split", "skinny_split8", "skinny_rc", "skinny_rc_jb",
                          "skinny_rc_multi", "skinny_flat", "hoist_vproj", "attn_lds", "flat_half", "half_min_mts", "gemm_x3_dma", "flat_xcd", "attn_skip0", "trunk_chain", "frontend_solo"))

ST_K, ST_V, ST_CKEY, ST_CVAL, ST_ECELL, ST_H, ST_C, ST_ENC, ST_STOPC = range(9)

_lib = None
_diag = None
_process_defaults: Dict[str, int] = {}
_vp, _i, _i64, _fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p


def _load(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(
            f"native library not built: {path} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C lip2speech_amd/csrc` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the Lip2Speech hot path.")
    L = ctypes.CDLL(path)
    _bind(L)
    if hasattr(L, "l2s_op_gemm"):
        _bind_diag(L)
    return L


def lib() -> ctypes.CDLL:
    """Load the native library (once).  Fails loudly - the HIP path is the only path."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def diag() -> ctypes.CDLL:
    """The diagnostic build (libl2s_diag.so, include/l2s_diag.h): the product ABI plus operator-level hooks, chain microbenches, stamped kernel builds
    and the block-form A/B options.  Loaded on demand by tools/ and the operator tests - never by the package's own callers."""
    global _diag
    if _diag is None:
        _diag = lib() if hasattr(lib(), "l2s_op_gemm") else _load(DIAG_LIB_PATH)
        if _diag is not _lib:
            for name, value in _process_defaults.items():      # the process defaults set so far hold for models of either library
                check(_diag.l2s_set_option(name.encode(), value), _diag)
    return _diag


def _bind(L: ctypes.CDLL) -> None:
    L.l2s_last_error.restype = ctypes.c_char_p
    L.l2s_abi_version.restype = _i
    L.l2s_model_create.argtypes = [ctypes.POINTER(_vp)]
    L.l2s_model_set_tensor.argtypes = [_vp, ctypes.c_char_p, _vp, _i64]
    L.l2s_model_finalize.argtypes = [_vp, _vp]
    L.l2s_model_destroy.argtypes = [_vp]
    L.l2s_min_T.argtypes = [_i]
    L.l2s_workspace_bytes.argtypes = [_i] * 5
    L.l2s_workspace_bytes.restype = _i64
    L.l2s_state_floats.argtypes = [_i, _i]
    L.l2s_state_floats.restype = _i64
    L.l2s_state_offset.argtypes = [_i, _i, _i]
    L.l2s_state_offset.restype = _i64
    L.l2s_encoder_fwd.argtypes = [_vp, _fp, _i, _i, _i, _i, _fp, _vp, _i64, _vp]
    L.l2s_normalise_pad_frames.argtypes = [_vp, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_int32), _i, _i, _i, _i, _fp, _vp]
    L.l2s_build_visual.argtypes = [_fp, _fp, _i, _i, _fp, _vp]
    L.l2s_decoder_prologue.argtypes = [_vp, _fp, _fp, _fp, _i, _i, _fp, _fp, _vp, _i64, _vp]
    L.l2s_decode_steps.argtypes = [_vp, _fp, _i, _i, _i, _fp, _vp, _fp, _fp, _fp, _i, _vp, _i64, _vp]
    L.l2s_postnet.argtypes = [_vp, _fp, _i, _i, _fp, _fp, _vp, _i64, _vp]
    L.l2s_output_lengths.argtypes = [_fp, _i, _i, _vp, _vp]
    L.l2s_speaker_workspace_bytes.argtypes = [_i, _i]
    L.l2s_speaker_workspace_bytes.restype = _i64
    L.l2s_speaker_encoder_fwd.argtypes = [_vp, _fp, _i, _i, _fp, _vp, _i64, _vp]
    L.l2s_inference.argtypes = [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _vp, _fp, _vp, _i64, _vp]
    L.l2s_workspace_bytes_multi.argtypes = [_i] * 6
    L.l2s_workspace_bytes_multi.restype = _i64
    L.l2s_inference_multi.argtypes = [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _fp, _vp, _fp, _vp, _i64, _vp]
    L.l2s_forward_eval.argtypes = [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _vp, _fp, _fp, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_forward_eval_multi.argtypes = [_vp, _i, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp,
                                         _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_model_set_option.argtypes = [_vp, ctypes.c_char_p, _i]
    L.l2s_set_option.argtypes = [ctypes.c_char_p, _i]
    L.l2s_train_bind.argtypes = [_vp, ctypes.c_char_p, _fp, _fp]
    L.l2s_train_postnet_tape_floats.argtypes = [_i, _i]
    L.l2s_train_postnet_tape_floats.restype = _i64
    L.l2s_train_postnet_ws_bytes.argtypes = [_i, _i]
    L.l2s_train_postnet_ws_bytes.restype = _i64
    L.l2s_train_postnet_fwd.argtypes = [_vp, _fp, _i, _i, _fp, _fp, _fp, _vp]
    L.l2s_train_postnet_bwd.argtypes = [_vp, _fp, _fp, _i, _i, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_train_steps_tape_floats.argtypes = [_i, _i]
    L.l2s_train_steps_tape_floats.restype = _i64
    L.l2s_train_steps_weights_floats.restype = _i64
    L.l2s_train_steps_ws_bytes.argtypes = [_i, _i]
    L.l2s_train_steps_ws_bytes.restype = _i64
    L.l2s_train_steps_pack_weights.argtypes = [_vp, _fp, _vp]
    L.l2s_train_steps_fwd.argtypes = [_vp, _fp, _i, _i, _i, _fp, _vp, _vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_train_steps_bwd.argtypes = [_vp, _fp, _i, _i, _i, _vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_train_set_bn.argtypes = [_vp, _i, ctypes.c_float]
    L.l2s_train_refresh_weights.argtypes = [_vp, _vp]
    L.l2s_train_encoder_tape_floats.argtypes = [_i, _i, _i]
    L.l2s_train_encoder_tape_floats.restype = _i64
    L.l2s_train_encoder_ws_bytes.argtypes = [_i, _i, _i]
    L.l2s_train_encoder_ws_bytes.restype = _i64
    L.l2s_train_encoder_fwd.argtypes = [_vp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _vp]
    L.l2s_train_encoder_bwd.argtypes = [_vp, _fp, _i, _i, _i, _i, _fp, _i, _fp, _vp, _i64, _vp]
    L.l2s_train_prologue_tape_floats.argtypes = [_i, _i]
    L.l2s_train_prologue_tape_floats.restype = _i64
    L.l2s_train_prologue_ws_bytes.argtypes = [_i, _i]
    L.l2s_train_prologue_ws_bytes.restype = _i64
    L.l2s_train_prologue_fwd.argtypes = [_vp, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_train_prologue_bwd.argtypes = [_vp, _fp, _fp, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _i64, _vp]
    L.l2s_train_scratch_bytes.restype = _i64
    L.l2s_loss.argtypes = [_fp] * 6 + [_i, _i, _i] + [_fp] * 5 + [_vp, _vp]
    L.l2s_grad_norm.argtypes = [_fp, _i64, _vp, _fp, _vp]
    _f = ctypes.c_float
    L.l2s_adamw_amsgrad_step.argtypes = [_fp] * 5 + [_i64, _f, _f, _f, _f, _f, _i, _fp, _f, _f, _vp]
    L.l2s_inverse_mel_workspace_bytes.argtypes = [_i] * 6
    L.l2s_inverse_mel_workspace_bytes.restype = _i64
    L.l2s_inverse_mel.argtypes = [_fp, _i, _fp, _i, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp, _vp, _i64, _vp]
    L.l2s_griffin_lim_workspace_bytes.argtypes = [_i, _i]
    L.l2s_griffin_lim_workspace_bytes.restype = _i64
    L.l2s_griffin_lim.argtypes = [_fp, _fp, _i, _i, _i, _i, _i, ctypes.c_float, _fp, _vp, _i64, _vp]
    L.l2s_estoi_workspace_bytes.argtypes = [_i]
    L.l2s_estoi_workspace_bytes.restype = _i64
    L.l2s_estoi.argtypes = [_fp, _fp, _i, _i, _fp, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), _fp, _vp, _i64, _vp]
    L.l2s_profile_enable.argtypes = [_i]
    L.l2s_profile_get.argtypes = [_i, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)]


def _bind_diag(L: ctypes.CDLL) -> None:
    L.l2s_op_gemm.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _vp]
    L.l2s_op_conv1d.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]
    L.l2s_op_gemm_ex.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _vp]
    L.l2s_op_conv1d_ex.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]
    L.l2s_op_conv1d_bwd.argtypes = [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _vp]
    L.l2s_op_frontend.argtypes = [_vp, _fp, _i, _i, _i, _i, _fp, _vp]
    L.l2s_op_launch_chain.argtypes = [_i, _i, _i, _i, _fp, _fp, _vp]
    L.l2s_op_skinny_timeline.argtypes = [_vp]
    L.l2s_op_attn_timeline.argtypes = [_vp]
    L.l2s_op_flat_timeline.argtypes = [_vp]
    L.l2s_op_pdecode_timeline.argtypes = [_vp, ctypes.c_int]
    L.l2s_op_gemm_x3_timeline.argtypes = [_vp, ctypes.c_int]
    L.l2s_op_fused_unit_timeline.argtypes = [_vp, _i]
    L.l2s_op_launch_chain2.argtypes = [_i, _i, _i, _i, _fp, _fp, _vp, _vp]
    L.l2s_op_step_attn_chain.argtypes = [_vp, _fp, _i, _i, _i, _vp, _i64, _vp]
    L.l2s_op_lstm_cell_chain.argtypes = [_vp, _i, _i, _vp, _i64, _vp, ctypes.POINTER(ctypes.c_double)]
    L.l2s_op_stamp_log.argtypes = [_vp, _i64]


def check(rc: int, L: Optional[ctypes.CDLL] = None) -> None:
    if rc != 0:
        raise RuntimeError("libl2s_hip: " + (L or lib()).l2s_last_error().decode("utf-8", "replace"))


def _dcheck(rc: int) -> None:
    check(rc, diag())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return t.data_ptr()


def _f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("the Lip2Speech hot path runs on the GPU: move inputs to cuda (no CPU fallback)")
    return t.detach().to(torch.float32).contiguous()


def min_T(T: int) -> int:
    return int(lib().l2s_min_T(T))


class NativeModel:
    """Owns an ``l2s_model`` (the packed device weight blob)."""

    def __init__(self, library: Optional[ctypes.CDLL] = None):
        self._L = library or lib()        # every call of this model goes to ONE library: the product, or the diagnostic build (tests of the A/B options, tools)
        self._h = _vp()
        self._check(self._L.l2s_model_create(ctypes.byref(self._h)))
        self._tls = threading.local()      # the workspace is per host thread: several threads may run batches on ONE model (one weight blob)
        self.calls = collections.Counter()  # whole-path entry points used, by C-ABI name (tests assert which route a caller took)

    def _check(self, rc: int) -> None:
        check(rc, self._L)

    def set_option(self, name: str, value: int) -> None:
        """Run-time option of THIS model (include/l2s.h "run-time options"); `native.set_option` changes the defaults of models created later."""
        self._check(self._L.l2s_model_set_option(self._h, name.encode(), int(value)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.l2s_model_destroy(self._h)
        except Exception:
            pass

    def load(self, tensors: Dict[str, torch.Tensor], keys: Iterable[str]) -> None:
        """Hand the checkpoint tensors named ``keys`` (reference key names) to the library and pack them."""
        L = self._L
        for key in keys:
            t = tensors[key]
            if not t.is_floating_point():
                continue                                   # num_batches_tracked: not used in eval arithmetic
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            self._check(L.l2s_model_set_tensor(self._h, key.encode(), a.ctypes.data_as(_vp), a.size))
        self._check(L.l2s_model_finalize(self._h, _stream()))

    # ------------------------------------------------------------------ workspace (caller-owned, cached)
    def workspace(self, B: int, T: int, H: int, W: int, S: int, device, G: int = 0) -> torch.Tensor:
        need = int(self._L.l2s_workspace_bytes_multi(G, B, T, H, W, S)) if G else int(self._L.l2s_workspace_bytes(B, T, H, W, S))
        ws = getattr(self._tls, "ws", None)
        if ws is None or ws.numel() < need or ws.device != device:
            ws = self._tls.ws = torch.empty(need, dtype=torch.uint8, device=device)
        return ws

    # ------------------------------------------------------------------ stages
    def encoder_fwd(self, video: torch.Tensor) -> torch.Tensor:
        video = _f32(video)
        B, C, T, H, W = video.shape
        assert C == 3, "frames are (B,3,T,H,W) RGB"
        feat = torch.empty(B, T, 768, dtype=torch.float32, device=video.device)
        ws = self.workspace(B, T, H, W, 1, video.device)
        self._check(self._L.l2s_encoder_fwd(self._h, _ptr(video), B, T, H, W, _ptr(feat), _ptr(ws), ws.numel(), _stream()))
        return feat

    def decoder_prologue(self, vis: torch.Tensor, emb: torch.Tensor, gumbel: torch.Tensor, want_dis: bool = True):
        vis, emb, gumbel = _f32(vis), _f32(emb), _f32(gumbel)
        B, T, C = vis.shape
        assert C == 1024 and emb.shape == (B, 256)
        m = min_T(T)
        assert m == 0 or gumbel.shape == (B * m, 501), f"gumbel noise must be {(B * m, 501)}"
        state = torch.zeros(int(self._L.l2s_state_floats(B, T)), dtype=torch.float32, device=vis.device)
        dis = torch.empty(B * m, 501, dtype=torch.float32, device=vis.device) if want_dis else None
        ws = self.workspace(B, T, 96, 96, 1, vis.device)
        self._check(self._L.l2s_decoder_prologue(self._h, _ptr(vis), _ptr(emb), _ptr(gumbel), B, T, _ptr(state), _ptr(dis),
                                         _ptr(ws), ws.numel(), _stream()))
        return state, dis

    def decode_steps(self, state: torch.Tensor, B: int, T: int, S: int, teacher: Optional[torch.Tensor] = None,
                     teacher_mask=None, want_attn: bool = True, attn_logits: bool = False):
        dev = state.device
        mel = torch.empty(B, S, 80, dtype=torch.float32, device=dev)
        stop = torch.empty(B, S, dtype=torch.float32, device=dev)
        attn = torch.empty(B, S, T, dtype=torch.float32, device=dev) if want_attn else None
        mask_buf = None
        if teacher is not None and teacher_mask is not None:
            teacher = _f32(teacher)
            assert teacher.shape == (B, S, 80)
            mask_np = np.ascontiguousarray(np.asarray(teacher_mask, dtype=np.uint8))
            assert mask_np.shape == (S,)
            mask_buf = mask_np.ctypes.data_as(_vp)
        else:
            teacher = None
        ws = self.workspace(B, T, 96, 96, S, dev)
        self._check(self._L.l2s_decode_steps(self._h, _ptr(state), B, T, S, _ptr(teacher), mask_buf, _ptr(mel), _ptr(stop),
                                     _ptr(attn), 1 if attn_logits else 0, _ptr(ws), ws.numel(), _stream()))
        return mel, stop, attn

    def postnet(self, mel: torch.Tensor, want_cf: bool = False):
        mel = _f32(mel)
        B, S, C = mel.shape
        assert C == 80
        out = torch.empty(B, 80, S, dtype=torch.float32, device=mel.device)
        cf = torch.empty(B, 80, S, dtype=torch.float32, device=mel.device) if want_cf else None
        ws = self.workspace(B, 29, 96, 96, S, mel.device)
        self._check(self._L.l2s_postnet(self._h, _ptr(mel), B, S, _ptr(out), _ptr(cf), _ptr(ws), ws.numel(), _stream()))
        return out, cf

    def forward_eval(self, video: torch.Tensor, emb: torch.Tensor, gumbel: torch.Tensor, S: int, teacher: Optional[torch.Tensor] = None,
                     teacher_mask=None):
        """`Lip2Speech.forward(..., tf_ratio)` in eval mode as ONE native call (`l2s_forward_eval`; what evaluate.py runs, evaluate.py:38):
        S = mels.shape[2] steps, free-running unless `teacher` (B,S,80) = cat(BOS, mels)[:, :S] and the per-step `teacher_mask` are given.
        Returns (mel (B,80,S), mel_post (B,80,S), stop (B,S), attention LOGITS (B,S,T), content_dis)."""
        return self.forward_eval_multi([(video, emb, gumbel, teacher)], S, teacher_mask=teacher_mask)[0]

    def forward_eval_multi(self, batches, S: int, teacher_mask=None):
        """Grouped `forward_eval` (`l2s_forward_eval_multi`): up to 8 (video, emb, gumbel[, teacher]) tuples of one shape through ONE launch
        chain; the batches share S and the scheduled-sampling mask.  Returns per batch (mel, mel_post, stop, attn_logits, content_dis) -
        views of one allocation, each bit-identical to `forward_eval` on that batch alone."""
        G = len(batches)
        assert 1 <= G <= 8, "1..8 batches per group"
        vids = [_f32(b[0]) for b in batches]
        embs = [_f32(b[1]) for b in batches]
        gums = [_f32(b[2]) for b in batches]
        B, _, T, H, W = vids[0].shape
        m = min_T(T)
        assert all(v.shape == vids[0].shape for v in vids) and all(e.shape == (B, 256) for e in embs), "the batches of a group share one shape"
        assert all(g.shape == (B * m, 501) for g in gums), f"gumbel noise must be {(B * m, 501)}"
        teach = mask_buf = mask_np = None
        if teacher_mask is not None and any(teacher_mask):
            mask_np = np.ascontiguousarray(np.asarray(teacher_mask, dtype=np.uint8))
            assert mask_np.shape == (S,)
            mask_buf = mask_np.ctypes.data_as(_vp)
            teach = [_f32(b[3]) for b in batches]
            assert all(t.shape == (B, S, 80) for t in teach), "teacher frames are (B,S,80) = cat(BOS, mels)[:, :S]"
        dev = vids[0].device
        mel_cf = torch.empty(G * B, 80, S, dtype=torch.float32, device=dev)
        mel_post = torch.empty(G * B, 80, S, dtype=torch.float32, device=dev)
        stop = torch.empty(G * B, S, dtype=torch.float32, device=dev)
        attn = torch.empty(G * B, S, T, dtype=torch.float32, device=dev)
        dis = torch.empty(G * B * m, 501, dtype=torch.float32, device=dev)
        arr = lambda ts: (_vp * G)(*[t.data_ptr() for t in ts])      # noqa: E731
        self.calls["l2s_forward_eval" if G == 1 else "l2s_forward_eval_multi"] += 1
        if G == 1:
            ws = self.workspace(B, T, H, W, S, dev)
            self._check(self._L.l2s_forward_eval(self._h, _ptr(vids[0]), _ptr(embs[0]), _ptr(gums[0]), B, T, H, W, S, _ptr(teach[0]) if teach else None, mask_buf,
                                         _ptr(mel_cf), _ptr(mel_post), _ptr(stop), _ptr(attn), _ptr(dis), _ptr(ws), ws.numel(), _stream()))
        else:
            ws = self.workspace(B, T, H, W, S, dev, G=G)
            self._check(self._L.l2s_forward_eval_multi(self._h, G, arr(vids), arr(embs), arr(gums), arr(teach) if teach else None, mask_buf, B, T, H, W, S,
                                               _ptr(mel_cf), _ptr(mel_post), _ptr(stop), _ptr(attn), _ptr(dis), _ptr(ws), ws.numel(), _stream()))
        return [(mel_cf[g * B:(g + 1) * B], mel_post[g * B:(g + 1) * B], stop[g * B:(g + 1) * B], attn[g * B:(g + 1) * B],
                 dis[g * B * m:(g + 1) * B * m]) for g in range(G)]

    def forward_eval_staged(self, video: torch.Tensor, emb: torch.Tensor, gumbel: torch.Tensor, S: int):
        """The same pass as five staged C-ABI calls (encoder, visual cat, prologue, loop, post-net) - the route `net.encoder` / `net.decoder`
        take when a caller uses them separately; the tests hold it bit-identical to `forward_eval`."""
        B, _, T, _, _ = video.shape
        feat = self.encoder_fwd(video)
        vis = build_visual(feat, emb)
        state, dis = self.decoder_prologue(vis, emb, gumbel)
        mel, stop, attn = self.decode_steps(state, B, T, S, want_attn=True, attn_logits=True)
        mel_post, mel_cf = self.postnet(mel, want_cf=True)
        return mel_cf, mel_post, stop, attn, dis

    def inference(self, video: torch.Tensor, emb: torch.Tensor, gumbel: torch.Tensor, S: int = 300, want_attn: bool = False):
        video, emb, gumbel = _f32(video), _f32(emb), _f32(gumbel)
        B, _, T, H, W = video.shape
        mel_post = torch.empty(B, 80, S, dtype=torch.float32, device=video.device)
        lengths = torch.empty(B, dtype=torch.int64, device=video.device)
        attn = torch.empty(B, S, T, dtype=torch.float32, device=video.device) if want_attn else None
        self.calls["l2s_inference"] += 1
        ws = self.workspace(B, T, H, W, S, video.device)
        self._check(self._L.l2s_inference(self._h, _ptr(video), _ptr(emb), _ptr(gumbel), B, T, H, W, S, _ptr(mel_post),
                                  _ptr(lengths), _ptr(attn), _ptr(ws), ws.numel(), _stream()))
        return mel_post, lengths, attn

    def inference_multi(self, batches, S: int = 300, want_attn: bool = False):
        """Grouped inference (`l2s_inference_multi`): `batches` = up to 8 (video, emb, gumbel) tuples of the same shape, advanced through ONE
        launch chain.  Returns [(mel_post (B,80,S), lengths (B,), attn (B,S,T) or None)] - views of one allocation, each bit-identical to
        `inference` on that batch."""
        G = len(batches)
        assert 1 <= G <= 8, "1..8 batches per group"
        vids = [_f32(b[0]) for b in batches]
        embs = [_f32(b[1]) for b in batches]
        gums = [_f32(b[2]) for b in batches]
        B, _, T, H, W = vids[0].shape
        assert all(v.shape == vids[0].shape for v in vids) and all(e.shape == (B, 256) for e in embs), "the batches of a group share one shape"
        dev = vids[0].device
        mel_post = torch.empty(G * B, 80, S, dtype=torch.float32, device=dev)
        lengths = torch.empty(G * B, dtype=torch.int64, device=dev)
        attn = torch.empty(G * B, S, T, dtype=torch.float32, device=dev) if want_attn else None
        self.calls["l2s_inference_multi"] += 1
        ws = self.workspace(B, T, H, W, S, dev, G=G)
        arr = lambda ts: (_vp * G)(*[t.data_ptr() for t in ts])      # noqa: E731
        self._check(self._L.l2s_inference_multi(self._h, G, arr(vids), arr(embs), arr(gums), B, T, H, W, S, _ptr(mel_post), _ptr(lengths), _ptr(attn),
                                        _ptr(ws), ws.numel(), _stream()))
        return [(mel_post[g * B:(g + 1) * B], lengths[g * B:(g + 1) * B], attn[g * B:(g + 1) * B] if want_attn else None) for g in range(G)]

    def speaker_encoder_fwd(self, audio: torch.Tensor) -> torch.Tensor:
        audio = _f32(audio)
        B, N = audio.shape
        emb = torch.empty(B, 256, dtype=torch.float32, device=audio.device)
        ws = torch.empty(int(self._L.l2s_speaker_workspace_bytes(B, N)), dtype=torch.uint8, device=audio.device)
        self._check(self._L.l2s_speaker_encoder_fwd(self._h, _ptr(audio), B, N, _ptr(emb), _ptr(ws), ws.numel(), _stream()))
        return emb

    # ------------------------------------------------------------------ training (forward with tapes + backward)
    def train_bind(self, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]) -> None:
        """Bind canonical-layout device parameters and their gradient slots by checkpoint key."""
        self._bound = (params, grads)                       # keep the tensors alive
        for key, p in params.items():
            if not p.is_floating_point():
                continue
            g = grads.get(key)
            self._check(self._L.l2s_train_bind(self._h, key.encode(), _ptr(p), _ptr(g) if g is not None else None))

    def train_set_bn(self, batch_stats: bool, momentum: float = 0.1) -> None:
        """BatchNorm of the training entry points: batch statistics + running-stat updates (nn.Module.train()) or running statistics."""
        self._check(self._L.l2s_train_set_bn(self._h, 1 if batch_stats else 0, float(momentum)))

    def train_refresh_weights(self) -> None:
        """Device-side re-pack of the weight blob from the bound tensors (needs self.set_option('refresh_map', 1) before load())."""
        self._check(self._L.l2s_train_refresh_weights(self._h, _stream()))

    def train_postnet_fwd(self, mel: torch.Tensor, drop: Optional[torch.Tensor] = None):
        """Post-net forward with a tape: mel (B,S,80) -> (mel_post (B,80,S), tape).  drop: packed dropout multipliers (postnet_drop_pack)."""
        mel = _f32(mel)
        B, S, _ = mel.shape
        L = self._L
        tape = torch.zeros(int(L.l2s_train_postnet_tape_floats(B, S)), dtype=torch.float32, device=mel.device)
        out = torch.empty(B, 80, S, dtype=torch.float32, device=mel.device)
        self._check(L.l2s_train_postnet_fwd(self._h, _ptr(mel), B, S, _ptr(tape), _ptr(out), _ptr(drop), _stream()))
        return out, tape

    def train_postnet_bwd(self, mel: torch.Tensor, dmel_post: torch.Tensor, tape: torch.Tensor, drop: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Post-net backward: dmel_post (B,80,S) -> dmel (B,S,80) (residual path included); parameter gradients into the bound slots."""
        mel, dmel_post = _f32(mel), _f32(dmel_post)
        B, S, _ = mel.shape
        L = self._L
        dmel = torch.zeros_like(mel)
        ws = torch.empty(int(L.l2s_train_postnet_ws_bytes(B, S)), dtype=torch.uint8, device=mel.device)
        self._check(L.l2s_train_postnet_bwd(self._h, _ptr(mel), _ptr(dmel_post), B, S, _ptr(tape), _ptr(dmel), _ptr(drop), _ptr(ws), ws.numel(), _stream()))
        return dmel

    def train_postnet(self, mel: torch.Tensor, dmel_post: torch.Tensor, drop: Optional[torch.Tensor] = None):
        """Post-net forward + backward (stage 1 of the training path): returns mel_post (B,80,S) and dmel (B,S,80)."""
        out, tape = self.train_postnet_fwd(mel, drop)
        return out, self.train_postnet_bwd(mel, dmel_post, tape, drop)

    def train_pack_weights(self, device) -> torch.Tensor:
        """Transposed step / prologue weights for the backward GEMMs, packed on the device from the bound parameters."""
        L = self._L
        wbuf = torch.empty(int(L.l2s_train_steps_weights_floats()), dtype=torch.float32, device=device)
        self._check(L.l2s_train_steps_pack_weights(self._h, _ptr(wbuf), _stream()))
        return wbuf

    def train_steps_fwd(self, state, B, T, S, teacher=None, teacher_mask=None, drop=None):
        """Loop forward with a tape.  Returns (mel (B,S,80), stop (B,S), attn_logits (B,S,T)) and the context for train_steps_bwd.
        drop: dict of dropout multipliers (any subset): 'prenet' (S,B,256), 'attn' (S,B,T), 'rnn' (S,B,512)."""
        drop = {k: _f32(v) for k, v in (drop or {}).items() if k in ("prenet", "attn", "rnn") and v is not None}
        dp, da, dr = drop.get("prenet"), drop.get("attn"), drop.get("rnn")
        assert dp is None or tuple(dp.shape) == (S, B, 256)
        assert da is None or tuple(da.shape) == (S, B, T)
        assert dr is None or tuple(dr.shape) == (S, B, 512)
        L, dev = self._L, state.device
        tape = torch.zeros(int(L.l2s_train_steps_tape_floats(B, S)), dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.l2s_train_steps_ws_bytes(B, S)), dtype=torch.uint8, device=dev)
        mel = torch.empty(B, S, 80, dtype=torch.float32, device=dev)
        stop = torch.empty(B, S, dtype=torch.float32, device=dev)
        logits = torch.empty(B, S, T, dtype=torch.float32, device=dev)
        mask_np = mask_dev = mask_ptr = None
        if teacher is not None:
            teacher = _f32(teacher)
            mask_np = np.ascontiguousarray(np.asarray(teacher_mask, dtype=np.uint8))
            mask_dev = torch.from_numpy(mask_np).to(dev)
            mask_ptr = mask_np.ctypes.data_as(_vp)
        self._check(L.l2s_train_steps_fwd(self._h, _ptr(state), B, T, S, _ptr(teacher), mask_ptr, _ptr(mask_dev) if mask_dev is not None else None,
                                    _ptr(tape), _ptr(mel), _ptr(stop), _ptr(logits), _ptr(dp), _ptr(da), _ptr(dr), _ptr(ws), ws.numel(), _stream()))
        ctx = dict(state=state, B=B, T=T, S=S, tape=tape, ws=ws, logits=logits, mask_np=mask_np, mask_ptr=mask_ptr, teacher=teacher, mask_dev=mask_dev,
                   drop=(dp, da, dr))
        return (mel, stop, logits), ctx

    def train_steps_bwd(self, ctx, dmel, dstop, wbuf=None):
        """BPTT through the loop: dmel (B,S,80), dstop (B,S) -> gradients of the prologue state; parameter gradients into the bound slots."""
        L, state = self._L, ctx["state"]
        B, T, S, dev = ctx["B"], ctx["T"], ctx["S"], state.device
        m = min_T(T)
        out = {"dk": torch.empty(B, T, 512, device=dev), "dv": torch.empty(B, T, 512, device=dev), "dckey": torch.empty(B, m, 256, device=dev),
               "dcval": torch.empty(B, m, 256, device=dev), "dh_init": torch.empty(2, B, 512, device=dev), "de_c": torch.empty(B, 512, device=dev)}
        if wbuf is None:
            wbuf = self.train_pack_weights(dev)
        ws = ctx["ws"]
        self._check(L.l2s_train_steps_bwd(self._h, _ptr(state), B, T, S, ctx["mask_ptr"], _ptr(ctx["tape"]), _ptr(ctx["logits"]), _ptr(_f32(dmel)), _ptr(_f32(dstop)),
                                    _ptr(wbuf), _ptr(out["dk"]), _ptr(out["dv"]), _ptr(out["dckey"]), _ptr(out["dcval"]), _ptr(out["dh_init"]), _ptr(out["de_c"]),
                                    _ptr(ctx["drop"][0]), _ptr(ctx["drop"][1]), _ptr(ctx["drop"][2]), _ptr(ws), ws.numel(), _stream()))
        return out

    def train_steps(self, state, B, T, S, dmel, dstop, teacher=None, teacher_mask=None, drop=None):
        """Loop forward-with-tape + BPTT (stage 2 of the training path).  Returns (mel, stop, attn_logits) and the state gradients."""
        outs, ctx = self.train_steps_fwd(state, B, T, S, teacher, teacher_mask, drop=drop)
        return outs, self.train_steps_bwd(ctx, dmel, dstop)

    def train_encoder_fwd(self, video, emb=None):
        """Encoder forward with a tape.  Returns (vis (B,T,1024) or None, feat (B,T,768), tape)."""
        video = _f32(video)
        B, _, T, H, W = video.shape
        L, dev = self._L, video.device
        tape = torch.empty(int(L.l2s_train_encoder_tape_floats(B, T, H)), dtype=torch.float32, device=dev)
        feat = torch.empty(B, T, 768, dtype=torch.float32, device=dev)
        vis = torch.empty(B, T, 1024, dtype=torch.float32, device=dev) if emb is not None else None
        self._check(L.l2s_train_encoder_fwd(self._h, _ptr(video), B, T, H, W, _ptr(_f32(emb)) if emb is not None else None, _ptr(vis), _ptr(feat), _ptr(tape), _stream()))
        return vis, feat, tape

    def train_encoder_bwd(self, video, dfeat, tape) -> None:
        """Encoder backward: dfeat (B,T,>=768) (e.g. dvis, row stride 1024); parameter gradients land in the bound slots."""
        video = _f32(video)
        B, _, T, H, W = video.shape
        L = self._L
        assert dfeat.is_cuda and dfeat.dtype == torch.float32 and dfeat.stride(-1) == 1 and dfeat.stride(0) == T * dfeat.stride(1)
        ws = torch.empty(int(L.l2s_train_encoder_ws_bytes(B, T, H)), dtype=torch.uint8, device=video.device)
        self._check(L.l2s_train_encoder_bwd(self._h, _ptr(video), B, T, H, W, dfeat.data_ptr(), int(dfeat.stride(1)), _ptr(tape), _ptr(ws), ws.numel(), _stream()))

    def train_prologue_fwd(self, vis, emb, gumbel):
        """Prologue forward with a tape (stage 3).  Returns (state, content_dis, tape)."""
        vis, emb, gumbel = _f32(vis), _f32(emb), _f32(gumbel)
        B, T, _ = vis.shape
        L, dev = self._L, vis.device
        state = torch.zeros(int(L.l2s_state_floats(B, T)), dtype=torch.float32, device=dev)
        dis = torch.empty(B * min_T(T), 501, dtype=torch.float32, device=dev)
        tape = torch.zeros(int(L.l2s_train_prologue_tape_floats(B, T)), dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.l2s_train_prologue_ws_bytes(B, T)), dtype=torch.uint8, device=dev)
        self._check(L.l2s_train_prologue_fwd(self._h, _ptr(vis), _ptr(emb), _ptr(gumbel), B, T, _ptr(state), _ptr(dis), _ptr(tape), _ptr(ws), ws.numel(), _stream()))
        return state, dis, tape

    def train_prologue_bwd(self, vis, emb, state, tape, g, dcontent_dis=None, wbuf=None):
        """Prologue backward: g = the state gradients of train_steps (dk, dv, dckey, dcval, dh_init, de_c).  Returns dvis (B,T,1024);
        parameter gradients land in the bound slots."""
        vis, emb = _f32(vis), _f32(emb)
        B, T, _ = vis.shape
        L, dev = self._L, vis.device
        if wbuf is None:
            wbuf = self.train_pack_weights(dev)
        ws = torch.empty(int(L.l2s_train_prologue_ws_bytes(B, T)), dtype=torch.uint8, device=dev)
        dvis = torch.empty(B, T, 1024, dtype=torch.float32, device=dev)
        gg = {k: _f32(v) for k, v in g.items()}
        dd = _f32(dcontent_dis) if dcontent_dis is not None else None
        self._check(L.l2s_train_prologue_bwd(self._h, _ptr(vis), _ptr(emb), B, T, _ptr(state), _ptr(tape), _ptr(wbuf), _ptr(gg["dk"]), _ptr(gg["dv"]),
                                       _ptr(gg["dckey"]), _ptr(gg["dcval"]), _ptr(gg["dh_init"]), _ptr(gg["de_c"]), _ptr(dd), _ptr(dvis),
                                       _ptr(ws), ws.numel(), _stream()))
        return dvis

    def lstm_cell_chain_us(self, B: int, n_pairs: int = 300) -> float:
        """Average duration of the decoder LSTM-cell kernel, one HIP-event pair around 2*n_pairs chained launches."""
        ws = self.workspace(B, 29, 96, 96, 300, torch.device("cuda", torch.cuda.current_device()))
        out = ctypes.c_double()
        D = diag()                          # diagnostic library; the model handle may come from either library
        check(D.l2s_op_lstm_cell_chain(self._h, B, n_pairs, _ptr(ws), ws.numel(), _stream(), ctypes.byref(out)), D)
        return float(out.value)

    def op_frontend(self, video: torch.Tensor) -> torch.Tensor:
        video = _f32(video)
        B, _, T, H, W = video.shape
        out = torch.empty(B * T, H // 4, W // 4, 24, dtype=torch.float32, device=video.device)
        D = diag()
        check(D.l2s_op_frontend(self._h, _ptr(video), B, T, H, W, _ptr(out), _stream()), D)
        return out


def postnet_drop_pack(masks) -> torch.Tensor:
    """Five post-net dropout multipliers in the reference's layout - (B,512,S) x4 and (B,80,S) - packed channel-last, back to back
    (the layout l2s_train_postnet_fwd/_bwd read)."""
    assert len(masks) == 5
    B, _, S = masks[0].shape
    out = torch.zeros(4 * B * S * 512 + B * S * 80, dtype=torch.float32, device=masks[0].device)
    for l, mk in enumerate(masks):
        C = 512 if l < 4 else 80
        assert tuple(mk.shape) == (B, C, S)
        out[l * B * S * 512: l * B * S * 512 + B * S * C] = mk.to(torch.float32).permute(0, 2, 1).reshape(-1)
    return out


def normalise_pad_frames(packed_u8: torch.Tensor, offsets, frames, H: int = 96, W: int = 96, T: Optional[int] = None) -> torch.Tensor:
    """Packed uint8 RGB clips on the device -> the model's (B,3,T,H,W) fp32 input (`l2s_normalise_pad_frames`): /255, ImageNet mean/std,
    zero padding to T (default: the longest clip).  `offsets` / `frames`: per-clip byte offsets into `packed_u8` and frame counts."""
    assert packed_u8.is_cuda and packed_u8.dtype == torch.uint8 and packed_u8.is_contiguous()
    B = len(frames)
    T = int(max(frames)) if T is None else int(T)
    video = torch.empty(B, 3, T, H, W, dtype=torch.float32, device=packed_u8.device)
    off = (_i64 * B)(*[int(o) for o in offsets])
    fr = (ctypes.c_int32 * B)(*[int(f) for f in frames])
    check(lib().l2s_normalise_pad_frames(packed_u8.data_ptr(), off, fr, B, T, H, W, _ptr(video), _stream()))
    return video


def build_visual(feat: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    feat, emb = _f32(feat), _f32(emb)
    B, T, _ = feat.shape
    vis = torch.empty(B, T, 1024, dtype=torch.float32, device=feat.device)
    check(lib().l2s_build_visual(_ptr(feat), _ptr(emb), B, T, _ptr(vis), _stream()))
    return vis


def output_lengths(stop: torch.Tensor) -> torch.Tensor:
    stop = _f32(stop)
    B, S = stop.shape
    out = torch.empty(B, dtype=torch.int64, device=stop.device)
    check(lib().l2s_output_lengths(_ptr(stop), B, S, _ptr(out), _stream()))
    return out


def state_field(state: torch.Tensor, B: int, T: int, field: int, shape) -> torch.Tensor:
    off = int(lib().l2s_state_offset(B, T, field))
    n = int(np.prod(shape))
    return state[off:off + n].view(*shape)


def op_gemm(A, Wt, scale=None, shift=None, actw=None, act: int = 0, x3: bool = False, bf16: bool = False, x3_narrow: bool = False,
            x3_dma: bool = False) -> torch.Tensor:
    A, Wt = _f32(A), _f32(Wt)
    M, K = A.shape
    N = Wt.shape[0]
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    _dcheck(diag().l2s_op_gemm_ex(_ptr(A), _ptr(Wt), _ptr(scale), _ptr(shift), _ptr(actw), _ptr(C), M, N, K, act, (1 if x3 else 0) | (2 if bf16 else 0) | (4 if x3_narrow else 0) | (8 if x3_dma else 0), _stream()))
    return C


def op_conv1d(X, Wp, scale=None, shift=None, actw=None, taps=1, stride=1, pad=0, act: int = 0, x3: bool = False, bf16: bool = False, x3_narrow: bool = False,
              x3_dma: bool = False) -> torch.Tensor:
    """X (B,Tin,Cin) channel-last, Wp (Cout, taps*Cin) tap-major."""
    X, Wp = _f32(X), _f32(Wp)
    B, Tin, Cin = X.shape
    Cout = Wp.shape[0]
    Tout = (Tin + 2 * pad - taps) // stride + 1
    out = torch.empty(B, Tout, Cout, dtype=torch.float32, device=X.device)
    _dcheck(diag().l2s_op_conv1d_ex(_ptr(X), _ptr(Wp), _ptr(scale), _ptr(shift), _ptr(actw), _ptr(out), B, Tin, Cin, Cout,
                                 taps, stride, pad, act, (1 if x3 else 0) | (2 if bf16 else 0) | (4 if x3_narrow else 0) | (8 if x3_dma else 0), _stream()))
    return out


def set_option(name: str, value: int) -> None:
    """Process DEFAULT of a run-time option: copied into models created afterwards (`NativeModel.set_option` changes one model)."""
    if name in DIAG_OPTIONS:
        _dcheck(diag().l2s_set_option(name.encode(), int(value)))      # a diagnostic switch: a default of diagnostic-library models only
        return
    check(lib().l2s_set_option(name.encode(), int(value)))
    _process_defaults[name] = int(value)
    if _diag is not None and _diag is not _lib:
        check(_diag.l2s_set_option(name.encode(), int(value)), _diag)


def set_thread_chains(n: int) -> None:
    """Tell the library how many launch chains the caller keeps in flight, for the calling host thread (include/l2s.h l2s_set_thread_chains): with
    two or more the step kernels take half-CU block forms so that chains overlap on the CUs.  A scheduling hint - results are the same bits."""
    check(lib().l2s_set_thread_chains(int(n)))
    if _diag is not None and _diag is not _lib:      # the hint is thread-local state of each library
        check(_diag.l2s_set_thread_chains(int(n)), _diag)


def persist_available() -> bool:
    """True where the persistent forms (option "persist_decode") can be taken on the current device: every workgroup resident at once (include/l2s.h)."""
    return bool(lib().l2s_persist_available())


def persist_timeouts() -> int:
    """Persistent launches of this process that gave up (their outputs are NaN).  Pinned host memory: meaningful after a synchronize."""
    return int(lib().l2s_persist_timeouts())


_timeouts_reported = 0


def check_persist_timeouts() -> None:
    """For a host that has just synchronized with the stream: raise if a persistent launch timed out since the last check (instead of handing its
    NaNs on).  Later persistent-eligible calls take the launch path (l2s_persist_available() is 0 from then on), so the error is raised once."""
    global _timeouts_reported
    n = persist_timeouts()
    if n > _timeouts_reported:
        new, _timeouts_reported = n - _timeouts_reported, n
        raise RuntimeError(f"{new} persistent decode launch(es) gave up after 2 s without progress (shared or CU-masked device): their outputs are NaN; "
                           f"later calls take the launch-per-phase path (or set option persist_decode=0)")


def op_conv1d_bwd(dZ, X, Wp, taps=1, stride=1, pad=0, want_dx=True):
    """Gradients of op_conv1d wrt its input (stride 1) and its tap-major weight."""
    dZ, X, Wp = _f32(dZ), _f32(X), _f32(Wp)
    B, Tin, Cin = X.shape
    Cout = Wp.shape[0]
    dX = torch.empty_like(X) if want_dx else None
    dW = torch.empty_like(Wp)
    _dcheck(diag().l2s_op_conv1d_bwd(_ptr(dZ), _ptr(X), _ptr(Wp), _ptr(dX), _ptr(dW), B, Tin, Cin, Cout, taps, stride, pad, _stream()))
    return dX, dW


# ---------------------------------------------------------------------------------------------- profiling
def profile_enable(on: bool) -> None:
    check(lib().l2s_profile_enable(1 if on else 0))


def profile_reset() -> None:
    check(lib().l2s_profile_reset())


def profile_read():
    """[(kernel name, launches, total ms)] measured with HIP events on the launch stream."""
    L = lib()
    out = []
    for i in range(L.l2s_profile_count()):
        name, n, ms = ctypes.c_char_p(), _i64(), ctypes.c_double()
        check(L.l2s_profile_get(i, ctypes.byref(name), ctypes.byref(n), ctypes.byref(ms)))
        out.append((name.value.decode(), int(n.value), float(ms.value)))
    return out


# ---------------------------------------------------------------------------------------------------------------- vocoder + metric (evaluate.py)
def inverse_mel(mel: torch.Tensor, fb: torch.Tensor, fb_nnz: int, init: torch.Tensor, iters: int, rows_per_call: Optional[int] = None,
                log_input: bool = False, want_loss: bool = False):
    """`l2s_inverse_mel`: mel (N, n_mels, L), fb (n_freqs, n_mels), init (N*L, n_freqs) -> spec (N, n_freqs, L)
    [, loss (calls, iters), iterations run per call (calls,) int32]."""
    mel, fb, init = _f32(mel), _f32(fb), _f32(init)
    N, M, Lf = mel.shape
    F = fb.shape[0]
    assert fb.shape[1] == M and tuple(init.shape) == (N * Lf, F), (fb.shape, init.shape)
    R = rows_per_call or N
    L = lib()
    ws = torch.empty(int(L.l2s_inverse_mel_workspace_bytes(N, Lf, M, F, R, iters)), dtype=torch.uint8, device=mel.device)
    spec = torch.empty(N, F, Lf, dtype=torch.float32, device=mel.device)
    loss = torch.empty(N // R, max(iters, 1), dtype=torch.float32, device=mel.device) if want_loss else None
    ran = torch.empty(N // R, dtype=torch.int32, device=mel.device) if want_loss else None
    check(L.l2s_inverse_mel(mel.data_ptr(), int(log_input), fb.data_ptr(), int(fb_nnz), init.data_ptr(), N, Lf, M, F, R, iters, spec.data_ptr(),
                            _ptr(loss), _ptr(ran), ws.data_ptr(), ws.numel(), _stream()))
    return (spec, loss, ran) if want_loss else spec


def griffin_lim(power_spec: torch.Tensor, init_angles: torch.Tensor, iters: int, n_fft: int = 1024, hop: int = 256, momentum: float = 0.99) -> torch.Tensor:
    """`l2s_griffin_lim`: power_spec (N, n_fft/2+1, L), init_angles complex (N, n_fft/2+1, L) or float (..., 2) -> wave (N, hop*(L-1))."""
    power_spec = _f32(power_spec)
    ang = torch.view_as_real(init_angles) if init_angles.is_complex() else init_angles
    ang = _f32(ang)
    N, F, Lf = power_spec.shape
    assert tuple(ang.shape) == (N, F, Lf, 2), ang.shape
    L = lib()
    ws = torch.empty(int(L.l2s_griffin_lim_workspace_bytes(N, Lf)), dtype=torch.uint8, device=power_spec.device)
    wave = torch.empty(N, hop * (Lf - 1), dtype=torch.float32, device=power_spec.device)
    check(L.l2s_griffin_lim(power_spec.data_ptr(), ang.data_ptr(), N, Lf, n_fft, hop, iters, momentum, wave.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return wave


def estoi(clean: torch.Tensor, pred: torch.Tensor, fir: Optional[torch.Tensor], up: int, down: int, n_pre_remove: int, n_resampled: int,
          band_lo_hi) -> torch.Tensor:
    """`l2s_estoi`: clean / pred (N, n_samples) on the device -> ESTOI per clip (N,).  `fir`: the padded polyphase filter (device) or None."""
    clean, pred = _f32(clean), _f32(pred)
    N, n = clean.shape
    assert pred.shape == clean.shape
    L = lib()
    bands = (ctypes.c_int * 30)(*[int(v) for v in band_lo_hi])
    ws = torch.empty(int(L.l2s_estoi_workspace_bytes(N)), dtype=torch.uint8, device=clean.device)
    score = torch.empty(N, dtype=torch.float32, device=clean.device)
    check(L.l2s_estoi(clean.data_ptr(), pred.data_ptr(), N, n, _ptr(fir), 0 if fir is None else fir.numel(), up, down, n_pre_remove, n_resampled,
                      bands, score.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return score
