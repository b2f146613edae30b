#!/usr/bin/env python
"""The one-launch GruBlock forward (csrc/gru_proj.hip) with parts switched off (LAB build: TPGSR_LAB=1 python -m tpgsr_amd.build; the
switch does not exist in a release build): what do the panel loads, the MFMA phase, the W_hh gather, the scan's stores and its gate
math cost on the SR trunk's geometry (N 48, 16 x 64, two-term arithmetic, the loaders the step uses)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

dev = "cuda"
N, H, W = 48, 16, 64
P = N * H * W
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
whh, bhh = R(2, 96, 32) / 32 ** 0.5, R(2, 96)
h, gates = torch.empty(P, 64, device=dev), torch.empty(P, 256, device=dev)
x, x2, strip = R(P, 64), R(P, 64), R(N * W, 32)
scale, shift = torch.ones(96, device=dev), torch.zeros(96, device=dev)
lib = _lib.load()


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


NOSCAN, NOPANEL, NOMFMA, NOWHH, NOSTORE, NOGATE = 1, 2, 4, 8, 16, 32
CASES = [(0, "everything"), (NOSCAN, "no scan: launch + panel + MFMA phase"), (NOSCAN | NOPANEL, "no scan, no panel loads: launch + MFMA phase"),
         (NOSCAN | NOPANEL | NOMFMA, "launch + barrier alone"), (NOWHH, "W_hh from constants"), (NOSTORE, "scan stores nothing"),
         (NOGATE, "gate math = two multiplies"), (NOPANEL | NOMFMA, "scan alone (W_hh gather, stores, gates)"),
         (NOPANEL | NOMFMA | NOWHH | NOSTORE, "scan alone, no W_hh gather, no stores"),
         (NOPANEL | NOMFMA | NOWHH | NOSTORE | NOGATE, "scan alone: LDS exchange + mat-vec only")]
with K.conv_terms(2):
    pas = {}
    for ax, Cin, kw in ((0, 64, dict(in2=x2)), (1, 96, dict(in_scale=scale, in_shift=shift, in_b=strip, cin_a=64))):
        wc, bc = R(Cin, 192) / Cin ** 0.5, R(192)
        K.make_bf_twin(wc, 0)
        geom = K.ConvGeom(N, H, W, Cin, 192)
        pas[ax] = (K.make_bigru_proj_args(K.make_conv_args(geom, x, wc, None, bias=bc, **kw), whh, bhh, ax, h, gates), wc, bc)
    print("| switched off | axis 0 (T 64) us | axis 1 (T 16) us |\n|---|---|---|")
    for bits, name in CASES:
        assert lib.tpgsr_gp_debug(bits) == 0
        print(f"| {name} | {timed(lambda: K.bigru_proj_fwd(pas[0][0])):.1f} | {timed(lambda: K.bigru_proj_fwd(pas[1][0])):.1f} |")
    lib.tpgsr_gp_debug(0)
