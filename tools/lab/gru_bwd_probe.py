#!/usr/bin/env python
"""bigru_bwd_kernel (csrc/gru.hip) with parts switched off (LAB build: TPGSR_LAB=1 python -m tpgsr_amd.build): what do the operand loads, the
stores and the LDS exchange + W_hh^T product cost per launch on the SR trunk's geometry (N 48, 16 x 64)?"""
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
gi, whh, bhh = R(P, 192), R(2, 96, 32) / 32 ** 0.5, R(2, 96)
h, gates = torch.empty(P, 64, device=dev), torch.empty(P, 256, device=dev)
dh, dgi, dghn = R(P, 64), torch.empty(P, 192, device=dev), torch.empty(P, 64, device=dev)
K.bigru_fwd(gi, whh, bhh, N, H, W, 0, h, gates)
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


print("| switched off | axis 0 (T 64) us | axis 1 (T 16) us |\n|---|---|---|")
for bits, name in [(0, "everything"), (8, "no time steps: launch + W_hh load"), (1, "operands from constants"), (2, "nothing stored"), (3, "no loads, no stores"),
                   (4, "no LDS exchange / W_hh^T product"), (7, "elementwise chain alone"), (3 | 0, "(again) no loads, no stores")]:
    assert lib.tpgsr_gru_debug(bits) == 0
    t = [timed(lambda: K.bigru_bwd2(gates, h, dh, None, whh, N, H, W, ax, dgi, dghn)) for ax in (0, 1)]
    print(f"| {name} | {t[0]:.1f} | {t[1]:.1f} |", flush=True)
lib.tpgsr_gru_debug(0)
