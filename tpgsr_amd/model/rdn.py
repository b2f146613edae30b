"""`--arch rdn_tl` (reference model/rdn.py:25-36 make_dense, :126-154 RDB_TL, :159-214 RDN_TL): residual dense network whose
three dense blocks fuse the text-prior map in their 1x1 bottleneck.  Same constructor / state_dict keys; executed operator by
operator on the HIP kernels (the dense concatenations are channel-slice copies, the x2 sub-pixel up-sampling is the conv
kernel's pixel-shuffle store)."""
import torch
from torch import nn

from .. import functional as Fh
from .nn_params import Conv2dParams, _NoForward
from .tl_common import InfoGen, spatial_text_embedding, zero_prior


class make_dense(nn.Module):
    def __init__(self, nChannels, growthRate, kernel_size=3):
        super().__init__()
        self.conv = Conv2dParams(nChannels, growthRate, kernel_size, padding=(kernel_size - 1) // 2, bias=False)

    def forward(self, x):
        return Fh.cat([x, Fh.relu(self.conv(x))])


class RDB_TL(nn.Module):
    def __init__(self, nChannels, nDenselayer, growthRate, out_text_channels=32):
        super().__init__()
        n = nChannels
        mods = []
        for _ in range(nDenselayer):
            mods.append(make_dense(n, growthRate))
            n += growthRate
        self.dense_layers = nn.Sequential(*mods)
        self.conv_1x1 = Conv2dParams(n + out_text_channels, nChannels, 1, padding=0, bias=False)

    def forward(self, x, text_emb):
        out = x
        for layer in self.dense_layers:
            out = layer(out)
        return Fh.add(self.conv_1x1(Fh.cat([out, text_emb])), x)


class sub_pixel(nn.Module):
    def __init__(self, scale, act=False):
        super().__init__()
        self.body = nn.Sequential(_NoForward())      # nn.PixelShuffle: fused into conv_up's store


class RDN_TL(nn.Module):
    def __init__(self, nChannel=4, nDenselayer=6, nFeat=64, scale_factor=2, growthRate=32, output_size=(32, 128), text_emb=37,
                 out_text_channels=32):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError("pixel-shuffle store specialised for scale 2")
        self.conv1 = Conv2dParams(nChannel, nFeat, 3, padding=1)
        self.conv2 = Conv2dParams(nFeat, nFeat, 3, padding=1)
        self.RDB1 = RDB_TL(nFeat, nDenselayer, growthRate, out_text_channels)
        self.RDB2 = RDB_TL(nFeat, nDenselayer, growthRate, out_text_channels)
        self.RDB3 = RDB_TL(nFeat, nDenselayer, growthRate, out_text_channels)
        self.GFF_1x1 = Conv2dParams(nFeat * 3, nFeat, 1, padding=0)
        self.GFF_3x3 = Conv2dParams(nFeat, nFeat, 3, padding=1)
        self.conv_up = Conv2dParams(nFeat, nFeat * scale_factor * scale_factor, 3, padding=1)
        self.upsample = sub_pixel(scale_factor)
        self.conv3 = Conv2dParams(nFeat, nChannel, 3, padding=1)
        self.tps_outputsize = [16, 64]
        self.infoGen = InfoGen(text_emb, out_text_channels)

    def forward(self, x, text_emb=None):
        if text_emb is None:
            text_emb = zero_prior(x, self.infoGen.tconv1.in_channels)
        t = spatial_text_embedding(self.infoGen, text_emb, (x.shape[2], x.shape[3]))
        F_ = self.conv1(Fh.to_nhwc(x))
        F_0 = self.conv2(F_)
        F_1 = self.RDB1(F_0, t)
        F_2 = self.RDB2(F_1, t)
        F_3 = self.RDB3(F_2, t)
        FGF = self.GFF_3x3(self.GFF_1x1(Fh.cat([F_1, F_2, F_3])))
        us = self.conv_up(Fh.add(FGF, F_), out_ps=True)
        return Fh.to_nchw(self.conv3(us))
