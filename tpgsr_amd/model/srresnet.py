"""`--arch srresnet_tl`: SRResNet with the text-prior strip concatenated into every residual block (reference:
model/srresnet.py:88-163 SRResNet_TL, :196-235 ResidualBlock_TL / UpsampleBLock) -- the conv + BN + PReLU backbone
`north_star` names.  Same constructor signature and state_dict keys; executed operator by operator on the HIP kernels
(tpgsr_amd/functional.py): MFMA convs with the pixel-shuffle store, fused BN(+act) passes, no ATen compute."""
import math

import torch
from torch import nn

from .. import functional as Fh
from .nn_params import BatchNormParams, Conv2dParams, PReLUParams, _NoForward
from .tl_common import InfoGen, spatial_text_embedding, zero_prior


class ResidualBlock_TL(nn.Module):
    def __init__(self, channels, out_text_channels=32):
        super().__init__()
        self.conv1 = Conv2dParams(channels, channels, 3, padding=1)
        self.bn1 = BatchNormParams(channels)
        self.prelu = PReLUParams()
        self.conv2 = Conv2dParams(channels + out_text_channels, channels, 3, padding=1)
        self.bn2 = BatchNormParams(channels)

    def forward(self, x, text_emb):
        """NHWC in / out"""
        r = self.prelu(self.bn1(self.conv1(x)))
        r = self.bn2(self.conv2(Fh.cat([r, text_emb])))
        return Fh.add(x, r)


class UpsampleBLock(nn.Module):
    """conv3x3 C -> 4C, PixelShuffle(2), PReLU (reference :224-235)"""

    def __init__(self, in_channels, up_scale):
        super().__init__()
        if up_scale != 2:
            raise NotImplementedError("pixel-shuffle store specialised for up_scale 2")
        self.conv = Conv2dParams(in_channels, in_channels * up_scale ** 2, 3, padding=1)
        self.pixel_shuffle = _NoForward()
        self.prelu = PReLUParams()

    def forward(self, x):
        return self.prelu(self.conv(x, out_ps=True))


class SRResNet_TL(nn.Module):
    def __init__(self, scale_factor=2, STN=False, width=128, height=32, mask=False, text_emb=37, out_text_channels=32):
        super().__init__()
        self.emb_cls = text_emb
        upsample_block_num = int(math.log(scale_factor, 2))
        in_planes = 4 if mask else 3
        self.block1 = nn.Sequential(Conv2dParams(in_planes, 64, 9, padding=4), PReLUParams())
        for i in range(2, 7):
            setattr(self, f"block{i}", ResidualBlock_TL(64, out_text_channels))
        self.block7 = nn.Sequential(Conv2dParams(64, 64, 3, padding=1), BatchNormParams(64))
        block8 = [UpsampleBLock(64, 2) for _ in range(upsample_block_num)]
        block8.append(Conv2dParams(64, in_planes, 9, padding=4))
        self.block8 = nn.Sequential(*block8)
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.tps_outputsize = [height // scale_factor, width // scale_factor]
        if STN:
            # the reference wires model/recognizer/stn_head.py here, whose five 2x2 max-pools need a >= 32-row input: on the
            # 16x64 TextZoom crops it fails in the reference too, and its training scripts never pass --STN for this arch
            raise NotImplementedError("SRResNet_TL is run without --STN (the reference's recognizer STN head cannot take 16x64 inputs)")
        self.stn = False
        self.infoGen = InfoGen(text_emb, out_text_channels)

    def forward(self, x, text_emb=None):
        if text_emb is None:
            text_emb = zero_prior(x, self.emb_cls)
        t = spatial_text_embedding(self.infoGen, text_emb, (x.shape[2], x.shape[3]))
        h = Fh.to_nhwc(x)
        b1 = self.block1[1](self.block1[0](h))
        b = b1
        for i in range(2, 7):
            b = getattr(self, f"block{i}")(b, t)
        b7 = self.block7[1](self.block7[0](b))
        out = Fh.add(b1, b7)
        for layer in self.block8:
            out = layer(out)
        return Fh.to_nchw(Fh.tanh(out))
