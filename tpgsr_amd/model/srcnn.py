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
