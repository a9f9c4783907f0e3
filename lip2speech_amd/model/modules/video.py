"""``VideoExtractor`` of the boundary (reference: /root/reference/model/modules/video.py:26-87).

Same constructor defaults, same ``state_dict`` keys (``frontend3D.*``, ``trunk.0.<unit>.banch*``,
``trunk.1.*``), same call contract ``(B,3,T,H,W) -> (B,T,768)`` L2-normalised; the arithmetic is
``l2s_encoder_fwd`` (fused Conv3d front-end + ShuffleNetV2 trunk in HIP).  Called on its own the module is
forward-only; gradients flow when it runs inside ``Lip2Speech.forward`` (``l2s_train_encoder_fwd/_bwd``, one autograd
node for the whole model).
"""
from __future__ import annotations

import torch

from ... import statespec
from ._tree import NativeBacked, ParamTree


class VideoExtractor(ParamTree, NativeBacked):
    _key_prefix = "encoder."

    def __init__(self, modality="video", hidden_dim=256, backbone_type="shufflenet", num_classes=500,
                 relu_type="prelu", tcn_options=None, width_mult=1.0, extract_feats=False):
        if backbone_type != "shufflenet" or width_mult != 1.0 or relu_type != "prelu":
            raise NotImplementedError("the HIP encoder implements the configuration the reference instantiates: "
                                      "ShuffleNetV2 1.0x trunk with a PReLU front-end (video.py:55-72)")
        ParamTree.__init__(self, statespec.encoder_spec(""), key_prefix="encoder.")
        self._init_native()
        self.frontend_nout = statespec.FRONT_CH
        self.backend_out = statespec.LAST_CH
        self.modality, self.backbone_type, self.extract_feats = modality, backbone_type, extract_feats

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise NotImplementedError("a stand-alone VideoExtractor is forward-only: train through Lip2Speech.forward (one autograd "
                                      "node over the HIP forward/backward of encoder + decoder) or call .eval() / torch.no_grad()")
        return self.native_model().encoder_fwd(x)
