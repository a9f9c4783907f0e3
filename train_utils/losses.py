"""Drop-in name ``train_utils.losses`` (reference: train_utils/losses.py:1-160).  ``from train_utils.losses import *`` in the reference's
train.py (train.py:26) also brings in the module's own imports - ``F`` (used at train.py:244), ``nn``, ``np``, ``torch`` - and ``AdversarialLoss``;
no ``__all__`` here, so the star import exports the same names."""
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn.functional as F  # noqa: F401
from torch import nn  # noqa: F401

from lip2speech_amd.losses import Loss, device  # noqa: F401


class AdversarialLoss(nn.Module):
    """train_utils/losses.py:83-160: the GAN terms over the mel discriminator.  The discriminator (model/modules/discriminator.py) is outside
    the hot path (OUT OF SCOPE; train.py constructs it only under its commented-out adversarial branch) - the name exists for
    the import surface and says so when constructed."""

    def __init__(self, optim_D=None):
        super().__init__()
        raise NotImplementedError("AdversarialLoss needs the mel discriminator, which is out of scope of the MI355X hot path (SURVEY.md section 2)")
