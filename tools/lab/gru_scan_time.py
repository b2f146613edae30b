#!/usr/bin/env python
"""The BiGRU kernels of one GruBlock alone on the SR trunk's geometry (N 48, 16 x 64): scan over a precomputed projection (tpgsr_bigru_fwd),
the one-launch forward (tpgsr_bigru_proj_fwd, two-term arithmetic, the loaders the step uses) and back-propagation through time
(tpgsr_bigru_bwd2) -- us per launch, both scan axes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tpgsr_amd import kernels as K  # noqa: E402

dev = "cuda"
N, H, W = 48, 16, 64
P = N * H * W
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
gi, whh, bhh = R(P, 192), R(2, 96, 32) / 32 ** 0.5, R(2, 96)
h, gates = torch.empty(P, 64, device=dev), torch.empty(P, 256, device=dev)
dh, dgi, dghn = R(P, 64), torch.empty(P, 192, device=dev), torch.empty(P, 64, device=dev)
x, x2, strip = R(P, 64), R(P, 64), R(N * W, 32)
scale, shift = torch.ones(96, device=dev), torch.zeros(96, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


print("| kernel | axis 0 (T 64, 768 sequences) us | axis 1 (T 16, 3072 sequences) us |\n|---|---|---|")
row = lambda name, f: print(f"| {name} | {timed(lambda: f(0)):.1f} | {timed(lambda: f(1)):.1f} |")
row("scan over a precomputed projection (bigru_fwd)", lambda ax: K.bigru_fwd(gi, whh, bhh, N, H, W, ax, h, gates))
with K.conv_terms(2):
    pas = {}
    for ax, Cin, kw in ((0, 64, dict(in2=x2)), (1, 96, dict(in_scale=scale, in_shift=shift, in_b=strip, cin_a=64))):
        wc, bc = R(Cin, 192) / Cin ** 0.5, R(192)
        K.make_bf_twin(wc, 0)
        geom = K.ConvGeom(N, H, W, Cin, 192)
        pas[ax] = (K.make_bigru_proj_args(K.make_conv_args(geom, x, wc, None, bias=bc, **kw), whh, bhh, ax, h, gates), wc, bc)
        assert K.bigru_proj_supported(pas[ax][0])
    row("projection + scan in one launch (bigru_proj_fwd, x2)", lambda ax: K.bigru_proj_fwd(pas[ax][0]))
K.bigru_fwd(gi, whh, bhh, N, H, W, 0, h, gates)
row("back-propagation through time (bigru_bwd2)", lambda ax: K.bigru_bwd2(gates, h, dh, None, whh, N, H, W, ax, dgi, dghn))
