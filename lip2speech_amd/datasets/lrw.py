"""LRW clip loader of the data boundary (reference: /root/reference/datasets/lrw/dataset.py:20-146).

On-disk format: ``<root>/LRW_Faces/<WORD>/<split>/<WORD>_<id>_{mouth,face}.npz`` are
bz2-compressed pickles of a list of JPEG byte arrays (mouth 96x96 RGB), ``<root>/lipread_audio/.../<id>.npz`` is a
numpy archive with key ``data`` (float32, 16 kHz).  JPEGs are decoded with PIL (the reference uses cv2; decoders
may differ by +-1 LSB, so real-data runs are plumbing checks, not parity gates).  The index CSV
``lrw500_detected_face.csv`` is missing from the reference's sample tree; when absent the index is rebuilt by
listing the mouth files.
"""
from __future__ import annotations

import bz2
import glob
import io
import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from .spectrograms import MelSpectrogram

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def load_frames(path: str) -> np.ndarray:
    """-> uint8 (T, H, W, 3) RGB"""
    from PIL import Image
    with bz2.BZ2File(path, "r") as f:
        blobs = pickle.load(f)
    return np.stack([np.asarray(Image.open(io.BytesIO(np.asarray(b).tobytes())).convert("RGB")) for b in blobs])


def normalise_mouth(frames_u8: np.ndarray) -> torch.Tensor:
    """uint8 (T,H,W,3) -> float32 (T,3,H,W), /255 then ImageNet mean/std (dataset.py:83-86)."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).permute(0, 3, 1, 2).float() / 255.0
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (x - mean) / std


def face_recog_resize(frame_u8: torch.Tensor) -> torch.Tensor:
    """dataset.py:76-79: `transforms.Resize((160, 160))` on a uint8 (3,H,W) tensor - bilinear, half-pixel centres, rounded back to uint8
    (torchvision's tensor path; the package is absent here, its published behaviour is restated) - then `(x - 127.5) / 128`."""
    x = torch.nn.functional.interpolate(frame_u8[None].float(), size=(160, 160), mode="bilinear", align_corners=False)[0]
    return (x.round().clamp(0, 255).to(torch.uint8).float() - 127.5) / 128.0


class LRW(Dataset):
    def __init__(self, rootpth, face_size=(96, 96), mode="train", demo=False, duration=1, face_augmentation=None, *args, raw_frames=False, **kwargs):
        """`raw_frames=True` (an extension): items carry the decoded uint8 clip `(T,H,W,3)` instead of the normalised fp32 one - the
        normalisation then runs on the device (`datasets.device.device_collate_fn_pad` + `PackedFrames.to_device`).  Extra positional /
        keyword arguments are accepted and ignored (the reference forwards them to `Dataset.__init__`, which takes none)."""
        super().__init__()
        self.raw_frames = raw_frames
        assert mode in ("train", "test", "val")
        self.rootpth, self.mode, self.demo = rootpth, mode, demo
        self.face_size, self.duration = face_size, duration
        # kept but never applied - exactly like the reference's LRW.__getitem__ (dataset.py:86-89 stores it, :123-146 does not call it)
        self.face_augmentation = face_augmentation if face_augmentation is not None else torch.nn.Identity()
        self.melspec_g = MelSpectrogram()
        index = os.path.join(rootpth, "lrw500_detected_face.csv")
        if os.path.exists(index):
            with open(index) as f:
                names = [l.split(",")[0] for l in f.read().splitlines() if l and l.split("/")[-2] == mode]
        else:
            pattern = os.path.join(rootpth, "LRW_Faces", "*", mode, "*_mouth.npz")
            names = sorted(os.path.relpath(p, os.path.join(rootpth, "LRW_Faces"))[:-len("_mouth.npz")] for p in glob.glob(pattern))
        self.items = [(os.path.join(rootpth, "LRW_Faces", f"{n}_face.npz"), os.path.join(rootpth, "LRW_Faces", f"{n}_mouth.npz"),
                       os.path.join(rootpth, "lipread_audio", f"{n}.npz")) for n in names]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        face_path, mouth_path, audio_path = self.items[idx]
        frames = load_frames(mouth_path)
        mouth = torch.from_numpy(np.ascontiguousarray(frames)) if self.raw_frames else normalise_mouth(frames)
        speech = torch.from_numpy(np.load(audio_path)["data"][np.newaxis])
        melspec = self.melspec_g(speech).squeeze(0)
        # two random face frames resized to 160x160 feed the third-party face tower (dataset.py:139-141).  The tower is outside this path,
        # but the draw is part of an epoch's RNG consumption: `torch.rand(2)` is taken exactly where the reference takes it, face file or not
        draw = torch.rand(2)
        face_crop = torch.zeros(2, 3, 160, 160)
        if os.path.exists(face_path):
            faces = torch.from_numpy(np.ascontiguousarray(load_frames(face_path))).permute(0, 3, 1, 2)
            face_crop = torch.stack([face_recog_resize(faces[int(i)]) for i in (draw * len(faces)).int()], dim=0)
        if self.demo:
            return mouth, speech, melspec, face_crop, (face_path, audio_path)
        return mouth, speech, melspec, face_crop
