"""Face speaker tower placeholder (reference: /root/reference/model/modules/vgg_face.py:12-60).

The reference wraps the third-party ``facenet_pytorch.InceptionResnetV1`` whose architecture and
weights are not part of the reference tree and whose constructor downloads a checkpoint; it is
outside the measured path ("parity unpinned") and is bypassed whenever a
``speaker_embedding`` is supplied (model.py:47-50).  The boundary keeps the attribute and the
``inference`` method so callers keep working; using it without the third-party package raises.

Checkpoints: a reference checkpoint carries the tower's tensors as ``vgg_face.resnet.*`` / ``vgg_face.projection_layer.*`` and
``demo.py:38`` loads it with ``strict=True``.  This module is an inert container for them: whatever keys a checkpoint holds under its
prefix are adopted on load (as buffers of nested sub-modules, same names, same values), so strict loading succeeds and
``state_dict()`` hands them back unchanged - a checkpoint saved here loads strictly in the reference again.
"""
import torch
from torch import nn


class _Held(nn.Module):
    """Nested name-space of adopted tensors (no arithmetic)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("inert container of third-party face-tower tensors")


class FaceRecognizer(nn.Module):
    def __init__(self):
        super().__init__()

    def _adopt(self, name: str, value: torch.Tensor):
        node = self
        parts = name.split(".")
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Held())
            node = node._modules[part]
        if parts[-1] not in node._buffers:
            node.register_buffer(parts[-1], value.detach().clone())

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for key, value in state_dict.items():
            if key.startswith(prefix) and isinstance(value, torch.Tensor):
                self._adopt(key[len(prefix):], value)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def inference(self, face_frames):
        raise RuntimeError("FaceRecognizer needs the third-party facenet_pytorch tower, which is outside this "
                           "path; pass speaker_embedding=... (the --encoding voice route, demo.py:84-86)")

    forward = inference
