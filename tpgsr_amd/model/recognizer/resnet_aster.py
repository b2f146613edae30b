"""ResNet_ASTER encoder (reference: model/recognizer/resnet_aster.py:35-131): layer0 conv3x3 + 5 stages of AsterBlocks
(conv1x1(stride) - BN - ReLU - conv3x3 - BN + shortcut - ReLU; 3, 4, 6, 6, 3 blocks; strides (2,2) (2,2) (2,1) (2,1) (2,1)) + a two-layer
bidirectional LSTM(512 -> 256).  Same state_dict keys / shapes / init.  Evaluation only."""
import math

import torch
from torch import nn

from ... import functional as Fh
from ..nn_params import BatchNormParams, Conv2dParams, _NoForward, _uniform


def conv3x3(in_planes, out_planes):
    return Conv2dParams(in_planes, out_planes, 3, padding=1, bias=False)


def conv1x1(in_planes, out_planes):
    return Conv2dParams(in_planes, out_planes, 1, padding=0, bias=False)


class AsterBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=(1, 1), downsample=None):
        super().__init__()
        self.conv1 = conv1x1(inplanes, planes)
        self.bn1 = BatchNormParams(planes)
        self.relu = _NoForward()
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNormParams(planes)
        self.downsample = downsample
        self.stride = tuple(stride)

    def forward(self, x):
        """x NHWC.  A strided 1x1 conv reads only every stride-th pixel: sub-sample first, then the dense 1x1 conv."""
        xs = Fh.subsample(x, self.stride[0], self.stride[1])
        out = self.bn1(self.conv1(xs), act="relu")
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample[1](self.downsample[0](xs))
        return Fh.relu(Fh.add(out, res))


class LSTM2Params(_NoForward):
    """nn.LSTM(input_size, hidden, bidirectional=True, num_layers=2, batch_first=True) parameters (default init)"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, 2
        bound = 1.0 / math.sqrt(hidden_size)
        g = 4 * hidden_size
        for layer in range(2):
            cin = input_size if layer == 0 else 2 * hidden_size
            for suf in ("", "_reverse"):
                self.register_parameter(f"weight_ih_l{layer}{suf}", nn.Parameter(_uniform(torch.empty(g, cin), bound)))
                self.register_parameter(f"weight_hh_l{layer}{suf}", nn.Parameter(_uniform(torch.empty(g, hidden_size), bound)))
                self.register_parameter(f"bias_ih_l{layer}{suf}", nn.Parameter(_uniform(torch.empty(g), bound)))
                self.register_parameter(f"bias_hh_l{layer}{suf}", nn.Parameter(_uniform(torch.empty(g), bound)))

    def flatten_parameters(self):
        pass

    def forward(self, x):
        """x (N, T, C) -> (N, T, 2 * hidden), evaluation only"""
        for layer in range(2):
            x = Fh.bilstm_eval(x, *[getattr(self, f"{nm}_l{layer}{suf}") for suf in ("", "_reverse")
                                    for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")])
        return x


class ResNet_ASTER(nn.Module):
    """For aster or crnn (reference docstring); with_lstm is whatever truthy value the caller passes (RecognizerBuilder hands over its
    `arch` string, recognizer_builder.py:44)"""

    def __init__(self, with_lstm=False, n_group=1):
        super().__init__()
        self.with_lstm, self.n_group = with_lstm, n_group
        self.layer0 = nn.Sequential(Conv2dParams(3, 32, 3, padding=1, bias=False), BatchNormParams(32), _NoForward())
        self.inplanes = 32
        self.layer1 = self._make_layer(32, 3, (2, 2))
        self.layer2 = self._make_layer(64, 4, (2, 2))
        self.layer3 = self._make_layer(128, 6, (2, 1))
        self.layer4 = self._make_layer(256, 6, (2, 1))
        self.layer5 = self._make_layer(512, 3, (2, 1))
        if with_lstm:
            self.rnn = LSTM2Params(512, 256)
            self.out_planes = 2 * 256
        else:
            self.out_planes = 512
        with torch.no_grad():      # reference :93-99: kaiming_normal_(fan_out, relu) for convs, BN (1, 0)
            for m in self.modules():
                if isinstance(m, Conv2dParams):
                    fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                    m.weight.normal_(0, math.sqrt(2.0 / fan_out))

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if tuple(stride) != (1, 1) or self.inplanes != planes:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes), BatchNormParams(planes))
        layers = [AsterBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(AsterBlock(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        """x (N, 3, 32, W) NCHW -> (N, W / 4, out_planes)"""
        if self.training:
            raise RuntimeError("ResNet_ASTER is an evaluation recognizer here (the reference never trains it); call .eval()")
        with torch.no_grad():
            h = Fh.to_nhwc(x)
            h = self.layer0[1](self.layer0[0](h), act="relu")
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5):
                for blk in layer:
                    h = blk(h)
            N, fh, fw, C = h.shape
            if fh != 1:
                raise ValueError(f"ResNet_ASTER expects 32-pixel-high inputs (feature map height {fh} after layer5)")
            feat = h.reshape(N, fw, C)                      # == x5.squeeze(2).transpose(2, 1)
            return self.rnn(feat) if self.with_lstm else feat
