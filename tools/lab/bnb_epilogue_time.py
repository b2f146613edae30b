#!/usr/bin/env python
"""Stand-alone timing of the trunk's data-gradient launch (N48 16x64 64->64 3x3, two-term) by epilogue: plain store, BatchNorm statistics,
BatchNorm-backward sums without / with the mish backward, each on the whole-CU halo kernel and on the two-workgroup halo kernel."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

DEV = "cuda"
lib = _lib.load()
N, H, W, Ci, Co = 48, 16, 64, 64, 64
g = torch.Generator().manual_seed(0)
x = torch.randn(N * H * W, Ci, generator=g).to(DEV)
w = (torch.randn(9 * Ci, Co, generator=g) / math.sqrt(9 * Ci)).to(DEV)
geom = K.ConvGeom(N, H, W, Ci, Co, 3, 3, 1, 1)
y = torch.randn(geom.M, Co, generator=g).to(DEV)
mean, rstd = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
bsc, bsh = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
out = torch.empty(geom.M, Co, device=DEV)
part = torch.empty((geom.M + 63) // 64, 2, Co, device=DEV)
cases = {
    "plain": {},
    "bn statistics": dict(bn_partial=part),
    "bnb, no activation": dict(bnb=dict(y=y, mean=mean, rstd=rstd, act="none", partial=part)),
    "bnb, mish backward": dict(bnb=dict(y=y, mean=mean, rstd=rstd, scale=bsc, shift=bsh, act="mish", partial=part)),
    "bnb, mish backward, store dz": dict(bnb=dict(y=y, mean=mean, rstd=rstd, scale=bsc, shift=bsh, act="mish", partial=part, store_dz=True)),
}
with K.conv_terms(2):
    K.make_bf_twin(w, Ci)
    for h3 in (1, 0):
        lib.tpgsr_halo3_set_enabled(h3)
        for name, kw in cases.items():
            try:
                args = K.make_conv_args(geom, x, w, out, **kw)
            except Exception as e:
                print(name, "->", type(e).__name__, e)
                continue
            for _ in range(5):
                K.conv_fwd(args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                K.conv_fwd(args)
            e1.record()
            torch.cuda.synchronize()
            print(f"halo3={h3}  {name:32s} {1e3 * e0.elapsed_time(e1) / 50:7.2f} us")
lib.tpgsr_halo3_set_enabled(1)
