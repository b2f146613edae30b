"""STN localisation head parameters (reference: model/stn_head.py:25-90): six conv3x3+BN+ReLU stages with max-pools,
fc1(512->512)+BN1d+ReLU, fc2(512->2*num_ctrlpoints) on 0.1*feat.  Same state_dict keys and the reference's custom
initialisation (conv N(0, sqrt(2/(k*k*Cout))), Linear N(0, 1e-3), fc2 = identity control points)."""
import math

import numpy as np
import torch
from torch import nn

from .. import functional as Fh
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

    POOLS = [(2, 2), (2, 2), (2, 2), (2, 2), (1, 2), None]     # MaxPool2d after stages 0-4 (reference :34-45)

    def forward(self, x):
        """(N, in_planes, 16, 64) -> (img_feat (N, 512), ctrl points (N, num_ctrlpoints, 2)) (reference :92-106).  Inside a
        TSRN the same computation is part of the fused plan; this standalone form runs it operator by operator."""
        h = Fh.to_nhwc(x)
        for i, pool in enumerate(self.POOLS):
            block = self.stn_convnet[2 * i]
            h = block[1](block[0](h), act="relu")
            if pool is not None:
                h = Fh.max_pool2d(h, pool, pool)
        N, fh, fw, C = h.shape
        fc1, bn1 = self.stn_fc1[0], self.stn_fc1[1]
        if fh * fw * C != fc1.in_features:
            raise ValueError(f"STNHead: the conv stack left a {fh}x{fw}x{C} map, stn_fc1 expects {fc1.in_features} features")
        # fc1 sees the NCHW flatten (c*fh*fw + y*fw + x) == a valid fh x fw conv over the NHWC map
        f = Fh.conv2d(h, fc1.weight.view(fc1.out_features, C, fh, fw), fc1.bias, 0)
        img_feat = bn1(f, act="relu")                                   # (N, 1, 1, 512)
        ctrl = Fh.conv2d(img_feat, self.stn_fc2.weight.view(self.stn_fc2.out_features, -1, 1, 1), self.stn_fc2.bias, 0,
                         wscale=0.1)                                     # fc2(0.1 * img_feat)
        return img_feat.reshape(N, -1), ctrl.reshape(N, self.num_ctrlpoints, 2)
