#!/usr/bin/env python
"""Launch the dominant kernel (3x3 64->64 implicit-GEMM conv at the C2 shape) a few times for rocprofv3 --pmc runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tpgsr_amd import kernels as K  # noqa: E402

N, H, W, C = 48, 16, 64, 64
g = K.ConvGeom(N, H, W, C, C, 3, 3, 1, 1)
x = torch.randn(g.M, C, device="cuda")
wf = torch.randn(g.K, C, device="cuda") * 0.05
out = torch.empty(g.M, C, device="cuda")
b = torch.randn(C, device="cuda")
a = K.make_conv_args(g, x, wf, out, bias=b)
junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for i in range(6):
    junk.fill_(i)            # flush the 256 MiB infinity cache between launches so HBM counters see cold reads
    K.conv_fwd(a)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", (g.M * C * 2 + g.K * C) * 4)
