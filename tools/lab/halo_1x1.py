#!/usr/bin/env python
"""1x1 convolutions (x3 arithmetic) through the tile loop vs through the halo kernel (tpgsr_halo_set_min_taps(1)): HIP-event
timing, interleaved rounds.  usage: halo_1x1.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
K.set_conv_prec("x3")
SHAPES = [("GRU proj dgrad 192->64", 48, 16, 64, 192, 64, False), ("GRU proj 64->192 (+residual)", 48, 16, 64, 64, 192, True),
          ("GRU proj 192->32", 48, 16, 64, 192, 32, False), ("LSTM in-proj 512->2048", 48, 1, 26, 512, 2048, False),
          ("LSTM in-proj 256->2048", 48, 1, 26, 256, 2048, False), ("LSTM dgrad 2048->512", 48, 1, 26, 2048, 512, False),
          ("embedding 512->256", 48, 1, 26, 512, 256, False)]
print("| shape | GFLOP | tile loop | halo kernel |\n|---|---|---|---|")
for name, N, H, W, Ci, Co, resid in SHAPES:
    g = K.ConvGeom(N, H, W, Ci, Co, 1, 1, 0, 0)
    x = torch.randn(g.M, Ci, device="cuda")
    x2 = torch.randn(g.M, Ci, device="cuda") if resid else None
    wf = torch.randn(g.K, Co, device="cuda") * 0.05
    out = torch.empty(g.M, Co, device="cuda")
    K.make_bf_twin(wf, Ci)
    a = K.make_conv_args(g, x, wf, out, in2=x2)
    res = {1: [], 2: []}
    for mt in (2, 1):
        lib.tpgsr_halo_set_min_taps(mt)
        K.conv_fwd(a)
    torch.cuda.synchronize()
    for _ in range(7):
        for mt in (2, 1):
            lib.tpgsr_halo_set_min_taps(mt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                K.conv_fwd(a)
            e1.record()
            torch.cuda.synchronize()
            res[mt].append(100.0 * e0.elapsed_time(e1))
    fl = 2.0 * g.M * g.K * Co
    t2, t1 = statistics.median(res[2]), statistics.median(res[1])
    print(f"| {name} | {fl / 1e9:.2f} | {t2:.1f} us {fl / t2 / 1e6:.0f} TF | {t1:.1f} us {fl / t1 / 1e6:.0f} TF |", flush=True)
lib.tpgsr_halo_set_min_taps(2)
