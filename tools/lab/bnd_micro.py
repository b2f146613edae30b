#!/usr/bin/env python
"""BatchNorm finalized by its consumer (csrc/bn_derive.h) against the two launches it replaces, alone on the chip: trunk shape
(M = 49152, C = 64) at 768 and 256 partial rows, forward (mish) and backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tpgsr_amd import kernels as K  # noqa: E402

DEV = "cuda"
M, C = 49152, 64
g = torch.Generator().manual_seed(0)
x, y, da = (torch.randn(M, C, generator=g).to(DEV) for _ in range(3))
one = torch.ones(C, device=DEV)
out = torch.empty(M, C, device=DEV)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for nrows in (768, 256, 64):
    rows = (torch.rand(nrows, 2, C, generator=g) * 100 + 1).to(DEV)
    rows[:, 1] += rows[:, 0] ** 2
    sc, sh, mu, rs = (torch.empty(C, device=DEV) for _ in range(4))
    coef = torch.empty(3, C, device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    dF = K.make_bn_derive(rows, nrows, C, M, one, bias=None, beta=one, running_mean=rm, running_var=rv, scale=sc, shift=sh, save_mean=mu, save_rstd=rs)
    K.bn_finalize(rows, nrows, C, M, None, one, one, rm, rv, sc, sh, mu, rs)
    dB = K.make_bn_derive(rows, nrows, C, M, one, save_mean=mu, save_rstd=rs, dgamma=dg, dbeta=db, coef=coef, accumulate=True)

    def fwd_pair():
        K.bn_finalize(rows, nrows, C, M, None, one, one, rm, rv, sc, sh, mu, rs)
        K.affine_act(x, M, C, sc, sh, "mish", out)

    def bwd_pair(act):
        K.bn_bwd_finalize(rows, nrows, C, M, one, mu, rs, dg, db, coef, accumulate=True)
        K.bn_bwd_apply(da, None, y, M, C, sc, sh, act, coef, out)

    print(f"rows {nrows}: forward  finalize + affine_act(mish) {timeit(fwd_pair):6.2f} us | affine_act alone {timeit(lambda: K.affine_act(x, M, C, sc, sh, 'mish', out)):6.2f} | "
          f"affine_act_bnd {timeit(lambda: K.affine_act_bnd(dF, x, M, 'mish', out)):6.2f}")
    for act in ("none", "mish"):
        print(f"rows {nrows}: backward finalize + apply({act}) {timeit(lambda: bwd_pair(act)):6.2f} us | apply alone "
              f"{timeit(lambda: K.bn_bwd_apply(da, None, y, M, C, sc, sh, act, coef, out)):6.2f} | bn_bwd_apply_bnd {timeit(lambda: K.bn_bwd_apply_bnd(dB, da, None, y, M, sc, sh, act, out)):6.2f}")
