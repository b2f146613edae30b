"""ResNet feature extractor of the `--tpg OPT` text-prior generator (reference
model/crnn/modules/feature_extraction.py:54-63 ResNet_FeatureExtractor, :151-193 BasicBlock, :196-246 ResNet [1, 2, 5, 3]).
Same constructor / state_dict keys; NHWC inside, operator by operator on the HIP kernels (MFMA convs, fused BN + ReLU passes,
max-pools; the one strided conv, conv4_1, runs as the stride-1 conv followed by row sub-sampling)."""
from torch import nn

from .... import functional as Fh
from ...nn_params import BatchNormParams, Conv2dParams, _NoForward


def _conv(cin, cout, k, pad):
    return Conv2dParams(cin, cout, k, padding=pad, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("the OPT feature extractor only uses stride-1 blocks")
        self.conv1 = _conv(inplanes, planes, 3, 1)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = _conv(planes, planes, 3, 1)
        self.bn2 = BatchNormParams(planes)
        self.relu = _NoForward()
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        x, xr = Fh.fork(x)       # two consumers: their gradients are summed by a HIP kernel, not by autograd
        out = self.bn1(self.conv1(x), act="relu")
        out = self.bn2(self.conv2(out))
        residual = xr if self.downsample is None else self.downsample[1](self.downsample[0](xr))
        return Fh.relu(Fh.add(out, residual))


class ResNet(nn.Module):
    def __init__(self, input_channel, output_channel, block, layers):
        super().__init__()
        self.output_channel_block = [int(output_channel / 4), int(output_channel / 2), output_channel, output_channel]
        ocb = self.output_channel_block
        self.inplanes = int(output_channel / 8)
        self.conv0_1 = _conv(input_channel, int(output_channel / 16), 3, 1)
        self.bn0_1 = BatchNormParams(int(output_channel / 16))
        self.conv0_2 = _conv(int(output_channel / 16), self.inplanes, 3, 1)
        self.bn0_2 = BatchNormParams(self.inplanes)
        self.relu = _NoForward()
        self.maxpool1 = _NoForward()
        self.layer1 = self._make_layer(block, ocb[0], layers[0])
        self.conv1 = _conv(ocb[0], ocb[0], 3, 1)
        self.bn1 = BatchNormParams(ocb[0])
        self.maxpool2 = _NoForward()
        self.layer2 = self._make_layer(block, ocb[1], layers[1], stride=1)
        self.conv2 = _conv(ocb[1], ocb[1], 3, 1)
        self.bn2 = BatchNormParams(ocb[1])
        self.maxpool3 = _NoForward()
        self.layer3 = self._make_layer(block, ocb[2], layers[2], stride=1)
        self.conv3 = _conv(ocb[2], ocb[2], 3, 1)
        self.bn3 = BatchNormParams(ocb[2])
        self.layer4 = self._make_layer(block, ocb[3], layers[3], stride=1)
        self.conv4_1 = _conv(ocb[3], ocb[3], 2, (0, 1))      # stride (2, 1): sub-sampled after the stride-1 conv
        self.bn4_1 = BatchNormParams(ocb[3])
        self.conv4_2 = _conv(ocb[3], ocb[3], 2, 0)
        self.bn4_2 = BatchNormParams(ocb[3])

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, 0), BatchNormParams(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        """NHWC in / out"""
        x = self.bn0_1(self.conv0_1(x), act="relu")
        x = self.bn0_2(self.conv0_2(x), act="relu")
        x = Fh.max_pool2d(x, 2, 2, 0)
        for blk in self.layer1:
            x = blk(x)
        x = self.bn1(self.conv1(x), act="relu")
        x = Fh.max_pool2d(x, 2, 2, 0)
        for blk in self.layer2:
            x = blk(x)
        x = self.bn2(self.conv2(x), act="relu")
        x = Fh.max_pool2d(x, 2, (2, 1), (0, 1))
        for blk in self.layer3:
            x = blk(x)
        x = self.bn3(self.conv3(x), act="relu")
        for blk in self.layer4:
            x = blk(x)
        x = self.bn4_1(Fh.subsample(self.conv4_1(x), 2, 1), act="relu")
        return self.bn4_2(self.conv4_2(x), act="relu")


class ResNet_FeatureExtractor(nn.Module):
    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        self.ConvNet = ResNet(input_channel, output_channel, BasicBlock, [1, 2, 5, 3])

    def forward(self, x):
        return self.ConvNet(x)
