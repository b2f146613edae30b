"""SRCNN (reference: model/srcnn.py:109-145) -- BASELINE config C1 only: a CPU plumbing case (SURVEY.md section 8a
row S1: "CPU, no kernel").  It is the one network of this package that is plain PyTorch, on purpose: C1 exists to
check the harness / fixtures / timer without a GPU, it is not part of the MI355X hot path."""
import torch
from torch import nn


class SRCNN(nn.Module):
    def __init__(self, scale_factor=2, in_planes=3, STN=False, height=32, width=128):
        super().__init__()
        if STN:
            raise NotImplementedError("SRCNN+STN is not on the TPGSR path (train_SRCNN.sh runs without --STN)")
        self.upscale_factor = scale_factor
        self.conv1 = nn.Conv2d(in_planes, 64, kernel_size=9, padding=4)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(64, 32, kernel_size=1, padding=0)
        self.relu2 = nn.ReLU()
        self.conv3 = nn.Conv2d(32, in_planes, kernel_size=5, padding=2)
        self.stn = STN

    def forward(self, x):
        x = torch.nn.functional.interpolate(x, scale_factor=self.upscale_factor)
        return self.conv3(self.relu2(self.conv2(self.relu1(self.conv1(x)))))


class SRCNN_TL(nn.Module):
    """`--arch srcnn_tl` (reference model/srcnn.py:50-106): SRCNN with the text-prior map concatenated in front of each conv.
    Unlike the CPU plumbing class above this one runs on the MI355X, operator by operator on the HIP kernels."""

    def __init__(self, scale_factor=2, in_planes=4, STN=False, height=32, width=128, text_emb=37, out_text_channels=32):
        super().__init__()
        from .nn_params import Conv2dParams
        from .tl_common import InfoGen
        self.upscale_factor = scale_factor
        self.conv1 = Conv2dParams(in_planes + out_text_channels, 64, 9, padding=4)
        self.relu1 = nn.Identity()
        self.conv2 = Conv2dParams(64 + out_text_channels, 32, 1, padding=0)
        self.relu2 = nn.Identity()
        self.conv3 = Conv2dParams(32 + out_text_channels, in_planes, 5, padding=2)
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.tps_outputsize = [height, width]
        if STN:   # as SRResNet_TL: the reference's recognizer STN head cannot take the 16x64 crops; its scripts run without --STN
            raise NotImplementedError("SRCNN_TL is run without --STN")
        self.stn = False
        self.infoGen = InfoGen(text_emb, out_text_channels)

    def forward(self, x, text_emb=None):
        from .. import functional as Fh
        from .tl_common import spatial_text_embedding, zero_prior
        if text_emb is None:
            text_emb = zero_prior(x, self.infoGen.tconv1.in_channels)
        h = Fh.upsample_nearest(Fh.to_nhwc(x), self.upscale_factor)
        t = spatial_text_embedding(self.infoGen, text_emb, (h.shape[1], h.shape[2]))
        out = Fh.relu(self.conv1(Fh.cat([h, t])))
        out = Fh.relu(self.conv2(Fh.cat([out, t])))
        return Fh.to_nchw(self.conv3(Fh.cat([out, t])))
