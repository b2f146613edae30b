#!/usr/bin/env python
"""BiGRU kernels alone: time per launch for the prefetch-ring depths (tpgsr_gru_set_prefetch) on the SR trunk's geometry (N 48, 16 x 64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

dev = "cuda"
N, H, W = 48, 16, 64
P = N * H * W
g = torch.Generator().manual_seed(0)
gi = torch.randn(P, 192, generator=g).to(dev)
whh = (torch.randn(2, 96, 32, generator=g) / 32 ** 0.5).to(dev)
bhh = torch.randn(2, 96, generator=g).to(dev)
h, gates = torch.empty(P, 64, device=dev), torch.empty(P, 256, device=dev)
dh = torch.randn(P, 64, generator=g).to(dev)
dgi, dgh = torch.empty(P, 192, device=dev), torch.empty(P, 192, device=dev)
lib = _lib.load()
print("| PF | axis | fwd us | bwd us |\n|---|---|---|---|")
for pf in (4, 8, 12):
    lib.tpgsr_gru_set_prefetch(pf)
    for axis in (0, 1):
        res = []
        for fn in (lambda: K.bigru_fwd(gi, whh, bhh, N, H, W, axis, h, gates), lambda: K.bigru_bwd(gates, h, dh, None, whh, N, H, W, axis, dgi, dgh)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(1e3 * e0.elapsed_time(e1) / 20)
        print(f"| {pf} | {axis} | {res[0]:.1f} | {res[1]:.1f} |")
