"""`--arch vdsr_tl` (reference model/vdsr.py:21-37 Conv_ReLU_Block_TL, :123-233 VDSR_TL): nearest x2 up-sampling, six
conv3x3(64 + 32 -> 64) + ReLU + skip blocks fed with the text-prior map, global residual.  Same constructor / state_dict keys /
initialisation (conv weights N(0, sqrt(2 / (k*k*Cout)))); executed operator by operator on the HIP kernels."""
from math import sqrt

import torch
from torch import nn

from .. import functional as Fh
from .nn_params import Conv2dParams
from .tl_common import InfoGen, spatial_text_embedding, zero_prior


class Conv_ReLU_Block_TL(nn.Module):
    def __init__(self, out_text_channels=32):
        super().__init__()
        self.conv = Conv2dParams(64 + out_text_channels, 64, 3, padding=1, bias=False)
        self.relu = nn.Identity()

    def forward(self, x, text_emb):
        """NHWC in / out"""
        return Fh.add(Fh.relu(self.conv(Fh.cat([x, text_emb]))), x)


class VDSR_TL(nn.Module):
    def __init__(self, scale_factor=2, in_planes=4, width=32, height=128, STN=False, text_emb=37, out_text_channels=32):
        super().__init__()
        self.upscale_factor = scale_factor
        self.out_text_channels = out_text_channels
        self.input = Conv2dParams(in_planes, 64, 3, padding=1, bias=False)
        self.output = Conv2dParams(64, in_planes, 3, padding=1, bias=False)
        self.relu = nn.Identity()
        self.infoGen = InfoGen(text_emb, out_text_channels)
        for i in range(1, 7):
            setattr(self, f"block{i}", Conv_ReLU_Block_TL(out_text_channels))
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, Conv2dParams):
                    n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                    m.weight.normal_(0, sqrt(2.0 / n))
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.tps_outputsize = [height, width]
        self.stn = False        # the reference hard-codes it off (:169)

    def forward(self, x, text_emb=None):
        if text_emb is None:
            text_emb = zero_prior(x, self.infoGen.tconv1.in_channels)
        h = Fh.upsample_nearest(Fh.to_nhwc(x), self.upscale_factor)
        t = spatial_text_embedding(self.infoGen, text_emb, tuple(self.tps_outputsize))
        out = Fh.relu(self.input(h))
        for i in range(1, 7):
            out = getattr(self, f"block{i}")(out, t)
        return Fh.to_nchw(Fh.add(self.output(out), h))
