"""Pieces shared by the `_TL` baseline backbones (reference: model/srresnet.py, model/srcnn.py, model/vdsr.py, model/rdn.py):
their InfoGen variant (four 2-D ConvTranspose2d + BN + ReLU: the text prior (N,37,1,26) grows to a (25, 213) map, unlike
TSRN's H = 1 strip) and the `F.interpolate(..., bilinear, align_corners=True)` to the image size.  Layer by layer on the HIP
kernels (tpgsr_amd/functional.py), NHWC inside."""
import torch
from torch import nn

from .. import functional as Fh
from .nn_params import BatchNormParams, ConvTranspose2dParams


class InfoGen(nn.Module):
    """reference model/srresnet.py:165-193 (identical copies in srcnn.py / vdsr.py / rdn.py)"""

    def __init__(self, t_emb, output_size):
        super().__init__()
        self.tconv1 = ConvTranspose2dParams(t_emb, 512, 3, 2, padding=0)
        self.bn1 = BatchNormParams(512)
        self.tconv2 = ConvTranspose2dParams(512, 128, 3, 2, padding=0)
        self.bn2 = BatchNormParams(128)
        self.tconv3 = ConvTranspose2dParams(128, 64, 3, 2, padding=1)
        self.bn3 = BatchNormParams(64)
        self.tconv4 = ConvTranspose2dParams(64, output_size, 3, (2, 1), padding=1)
        self.bn4 = BatchNormParams(output_size)

    def nhwc(self, t):
        x = self.bn1(self.tconv1(t), act="relu")
        x = self.bn2(self.tconv2(x), act="relu")
        x = self.bn3(self.tconv3(x), act="relu")
        return self.bn4(self.tconv4(x), act="relu")

    def forward(self, t_embedding):
        return Fh.to_nchw(self.nhwc(Fh.to_nhwc(t_embedding)))


def spatial_text_embedding(info_gen: InfoGen, text_emb: torch.Tensor, size):
    """infoGen(text_emb) resized to `size` = (H, W), NHWC (the reference's spatial_t_emb)"""
    return Fh.interpolate_bilinear(info_gen.nhwc(Fh.to_nhwc(text_emb)), size)


def zero_prior(x, emb_cls):
    return torch.zeros(x.shape[0], emb_cls, 1, 26, device=x.device)
