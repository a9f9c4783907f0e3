"""Drop-in name ``train_utils.losses`` (reference: train_utils/losses.py:13-79): ``from train_utils.losses import *`` gives ``Loss``."""
from lip2speech_amd.losses import Loss, device  # noqa: F401

__all__ = ["Loss", "device"]
