"""STN localisation head parameters (reference: model/stn_head.py:25-90): six conv3x3+BN+ReLU stages with max-pools,
fc1(512->512)+BN1d+ReLU, fc2(512->2*num_ctrlpoints) on 0.1*feat.  Same state_dict keys and the reference's custom
initialisation (conv N(0, sqrt(2/(k*k*Cout))), Linear N(0, 1e-3), fc2 = identity control points)."""
import math

import numpy as np
import torch
from torch import nn

from .nn_params import BatchNormParams, Conv2dParams, LinearParams, _NoForward


class _Slot(_NoForward):
    """placeholder keeping nn.Sequential indices identical to the reference (pool / ReLU layers hold no state)"""


def conv3x3_block(in_planes, out_planes):
    return nn.Sequential(Conv2dParams(in_planes, out_planes, 3, padding=1), BatchNormParams(out_planes), _Slot())


class STNHead(nn.Module):
    def __init__(self, in_planes, num_ctrlpoints, activation="none", input_size=(32, 128)):
        super().__init__()
        if activation != "none":
            raise NotImplementedError("only activation='none' is on the TPGSR path (model/tsrn.py:57,175)")
        self.in_planes, self.num_ctrlpoints, self.activation = in_planes, num_ctrlpoints, activation
        chans = [(in_planes, 32), (32, 64), (64, 128), (128, 256), (256, 256), (256, 256)]
        layers = []
        for i, (ci, co) in enumerate(chans):
            layers.append(conv3x3_block(ci, co))
            if i < 5:
                layers.append(_Slot())  # MaxPool2d slots (indices 1,3,5,7,9)
        self.stn_convnet = nn.Sequential(*layers)
        self.stn_fc1 = nn.Sequential(LinearParams(2 * 256, 512), BatchNormParams(512), _Slot())
        self.stn_fc2 = LinearParams(512, num_ctrlpoints * 2)
        self.init_weights(self.stn_convnet)
        self.init_weights(self.stn_fc1)
        self.init_stn(self.stn_fc2)

    @staticmethod
    def init_weights(module):
        with torch.no_grad():
            for m in module.modules():
                if isinstance(m, Conv2dParams):
                    n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                    m.weight.normal_(0, math.sqrt(2.0 / n))
                    m.bias.zero_()
                elif isinstance(m, BatchNormParams):
                    m.weight.fill_(1)
                    m.bias.zero_()
                elif isinstance(m, LinearParams):
                    m.weight.normal_(0, 0.001)
                    m.bias.zero_()

    def init_stn(self, fc2):
        margin = 0.01
        k = self.num_ctrlpoints // 2
        xs = np.linspace(margin, 1.0 - margin, k)
        pts = np.concatenate([np.stack([xs, np.full(k, margin)], 1), np.stack([xs, np.full(k, 1 - margin)], 1)], 0)
        with torch.no_grad():
            fc2.weight.zero_()
            fc2.bias.copy_(torch.tensor(pts.astype(np.float32)).view(-1))

    def forward(self, x):
        raise RuntimeError("STNHead is executed inside the fused TSRN plan; it has no standalone forward in tpgsr_amd")
