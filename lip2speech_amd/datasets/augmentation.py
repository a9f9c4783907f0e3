"""`FaceAugmentation` of the data boundary (reference: /root/reference/datasets/augmentation.py:8-19).

A coin flip per call: with probability ``p`` the frames pass through, otherwise every frame is mirrored left-right
(the reference calls torchvision's ``TF.hflip`` per frame; a flip of the last axis is the same operation on
``(..., H, W)`` tensors).  Note that the reference's LRW loader stores the augmentation and never applies it
(datasets/lrw/dataset.py:86-89,123-146); GRID/AVSpeech/WILD apply it to the mouth crops.
"""
from __future__ import annotations

import torch
from torch import nn


class FaceAugmentation(nn.Module):
    def __init__(self, p: float = 0.5):
        super().__init__()
        self.p = p

    def forward(self, faces):
        if torch.rand(1) < self.p:
            return faces
        return [torch.flip(torch.as_tensor(face), dims=(-1,)) for face in faces]
