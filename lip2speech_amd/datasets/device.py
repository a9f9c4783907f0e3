"""Device-side half of the data boundary (reference: datasets/lrw/dataset.py:83-86,123-146 and
datasets/__init__.py:7-46).

The reference normalises every decoded frame to fp32 on the CPU and its collate zero-pads and permutes the clips into the
`(B,3,T,96,96)` batch - 102.6 MB per B=32 LRW batch assembled by the host and copied over PCIe.  Here the loader hands over the decoded
uint8 frames (`LRW(..., raw_frames=True)`), `device_collate_fn_pad` packs them back to back into ONE pinned buffer (25.7 MB) next to the
usual audio / mel / gate padding, and `PackedFrames.to_device()` runs `l2s_normalise_pad_frames`: the same three fp32 operations per value
(`/255`, `- mean`, `/ std`), the zero padding and the layout change in one kernel - bit-identical to `train_collate_fn_pad` on the
normalised clips (tests/test_data_boundary.py, tests/test_gpu_parity.py).
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from .. import native


class PackedFrames:
    """B clips of uint8 RGB frames `(T_b,H,W,3)` packed back to back (4-byte aligned) in one host buffer."""

    def __init__(self, clips: Sequence[torch.Tensor], pin: bool = True):
        assert len(clips) > 0 and all(c.dtype == torch.uint8 and c.dim() == 4 and c.shape[3] == 3 for c in clips), "clips are uint8 (T,H,W,3)"
        self.H, self.W = int(clips[0].shape[1]), int(clips[0].shape[2])
        assert all(c.shape[1:3] == clips[0].shape[1:3] for c in clips), "one crop size per batch"
        self.frames: List[int] = [int(c.shape[0]) for c in clips]
        self.offsets: List[int] = []
        total = 0
        for c in clips:
            self.offsets.append(total)
            total += (c.numel() + 3) // 4 * 4
        buf = torch.empty(total, dtype=torch.uint8)
        if pin and torch.cuda.is_available():
            buf = buf.pin_memory()
        for c, o in zip(clips, self.offsets):
            buf[o:o + c.numel()] = c.reshape(-1)
        self.data = buf

    @property
    def lengths(self) -> torch.Tensor:
        return torch.tensor(self.frames)

    def to_device(self, device="cuda", T: int = None) -> torch.Tensor:
        """-> `(B,3,T,H,W)` fp32 on the device, normalised and zero-padded (T defaults to the longest clip)."""
        dev = self.data.to(device, non_blocking=True)
        return native.normalise_pad_frames(dev, self.offsets, self.frames, self.H, self.W, T)


def device_collate_fn_pad(batch):
    """`train_collate_fn_pad` for items whose first element is the RAW clip (uint8 `(T,H,W,3)`, `LRW(raw_frames=True)`): the same
    4-tuple, with `PackedFrames` in place of the padded fp32 video - call `.to_device()` on it where the reference's loop calls
    `videos.to(device)` (train.py:163)."""
    from . import _pad_audio_mels
    with_paths = len(batch[0]) == 5
    packed = PackedFrames([b[0] for b in batch])
    out = ((packed, packed.lengths),) + _pad_audio_mels([b[1] for b in batch], [b[2] for b in batch]) + (torch.stack([b[3] for b in batch], dim=0),)
    return out + (tuple(b[4] for b in batch),) if with_paths else out
