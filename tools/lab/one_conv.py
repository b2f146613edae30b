#!/usr/bin/env python
"""one conv shape / arithmetic mode, launched a few times (for rocprofv3 --pmc / --kernel-trace runs)
usage: one_conv.py N H W Cin Cout KH KW pad mode[f32|x3|bf16] [wgrad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tpgsr_amd import kernels as K  # noqa: E402

N, H, W, Ci, Co, KH, KW, pad = (int(v) for v in sys.argv[1:9])
mode = sys.argv[9]
wgrad = len(sys.argv) > 10
g = K.ConvGeom(N, H, W, Ci, Co, KH, KW, pad if KH > 1 else 0, pad if KW > 1 else 0)
x = torch.randn(g.N * H * W, Ci, device="cuda")
wf = torch.randn(g.K, Co, device="cuda") * 0.05
out = torch.empty(g.M, Co, device="cuda")
dy = torch.randn(g.M, Co, device="cuda")
K.make_bf_twin(wf, Ci)
K.set_conv_prec(mode)
if wgrad:
    Z = K.wgrad_splits(g.M, g.K, Co)
    part = torch.empty(Z, g.K, Co, device="cuda")
    a = K.make_wgrad_args(K.make_conv_args(g, x), dy, part, None)
    fn = lambda: K.conv_wgrad(a)
else:
    a = K.make_conv_args(g, x, wf, out)
    fn = lambda: K.conv_fwd(a)
for _ in range(8):
    fn()
torch.cuda.synchronize()
