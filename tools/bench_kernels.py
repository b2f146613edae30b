#!/usr/bin/env python
"""Micro-benchmarks of the hot kernels at the C2 shapes (bs 48, 16x64 LR): HIP-event timing on the launch stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tpgsr_amd import kernels as K  # noqa: E402

DEV = "cuda"
N, H, W = 48, 16, 64


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps  # us


def conv_case(name, Hh, Ww, Ci, Co, KH, KW, ph, pw, **kw):
    g = K.ConvGeom(N, Hh, Ww, Ci, Co, KH, KW, ph, pw)
    x = torch.randn(N * Hh * Ww, Ci, device=DEV)
    wf = torch.randn(g.K, Co, device=DEV) * 0.05
    out = torch.empty(g.M * (1 if not kw.get("out_ps") else 1), Co, device=DEV)
    b = torch.randn(Co, device=DEV)
    extra = {}
    if kw.get("prologue"):
        extra = dict(in_scale=torch.rand(Ci, device=DEV) + 0.5, in_shift=torch.randn(Ci, device=DEV), in_act="mish")
    part = torch.empty((g.M + 63) // 64, 2, Co, device=DEV) if kw.get("stats") else None
    a = K.make_conv_args(g, x, wf, out, bias=b, bn_partial=part, **extra)
    us = timeit(lambda: K.conv_fwd(a))
    fl = 2.0 * g.M * g.K * Co
    print(f"conv_fwd  {name:34s} {us:8.1f} us  {fl / us / 1e6:7.2f} TFLOP/s")
    dy = torch.randn(g.M, Co, device=DEV)
    Z = K.wgrad_splits(g.M, g.K, Co)
    partw = torch.empty(Z, g.K, Co, device=DEV)
    dbp = torch.empty(Z, Co, device=DEV)
    ca = K.make_conv_args(g, x, **extra)
    wa = K.make_wgrad_args(ca, dy, partw, dbp)
    us = timeit(lambda: K.conv_wgrad(wa))
    print(f"conv_wgrad {name:33s} {us:8.1f} us  {fl / us / 1e6:7.2f} TFLOP/s  (Z={Z})")
    dw = torch.zeros(Co * g.K, device=DEV)
    db = torch.zeros(Co, device=DEV)
    us = timeit(lambda: K.wgrad_reduce(partw, dbp, Z, g, dw, db))
    print(f"wgrad_reduce {name:31s} {us:8.1f} us  ({Z * g.K * Co * 4 / 1e6:.1f} MB partials)")


def gru_case(axis):
    P = N * H * W
    gi = torch.randn(P, 192, device=DEV)
    whh = torch.randn(2, 96, 32, device=DEV) * 0.1
    bhh = torch.randn(2, 96, device=DEV) * 0.1
    h = torch.empty(P, 64, device=DEV)
    gates = torch.empty(P, 256, device=DEV)
    us = timeit(lambda: K.bigru_fwd(gi, whh, bhh, N, H, W, axis, h, gates))
    print(f"bigru_fwd axis={axis}  {us:8.1f} us")
    dh = torch.randn(P, 64, device=DEV)
    dgi = torch.empty(P, 192, device=DEV)
    dgh = torch.empty(P, 192, device=DEV)
    us = timeit(lambda: K.bigru_bwd(gates, h, dh, None, whh, N, H, W, axis, dgi, dgh))
    print(f"bigru_bwd axis={axis}  {us:8.1f} us")


def bn_case():
    P, Cc = N * H * W, 64
    nblk = (P + 63) // 64
    part = torch.randn(nblk, 2, Cc, device=DEV).abs()
    gamma, beta = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV)
    rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    scale, shift, mean, rstd = (torch.empty(Cc, device=DEV) for _ in range(4))
    us = timeit(lambda: K.bn_finalize(part, nblk, Cc, P, None, gamma, beta, rm, rv, scale, shift, mean, rstd))
    print(f"bn_finalize C=64 nblk={nblk}  {us:8.1f} us")
    y, da, dy = torch.randn(P, Cc, device=DEV), torch.randn(P, Cc, device=DEV), torch.empty(P, Cc, device=DEV)
    nb = min(1024, P // 64)
    bp = torch.empty(nb, 2, Cc, device=DEV)
    coef = torch.empty(3, Cc, device=DEV)
    dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    us = timeit(lambda: K.bn_bwd_reduce(da, None, y, P, Cc, scale, shift, mean, rstd, "mish", bp, nb))
    print(f"bn_bwd_reduce (mish)  {us:8.1f} us")
    us = timeit(lambda: K.bn_bwd_finalize(bp, nb, Cc, P, gamma, mean, rstd, dg, db, coef))
    print(f"bn_bwd_finalize       {us:8.1f} us")
    us = timeit(lambda: K.bn_bwd_apply(da, None, y, P, Cc, scale, shift, "mish", coef, dy))
    print(f"bn_bwd_apply (mish)   {us:8.1f} us")
    a1 = torch.empty(P, Cc, device=DEV)
    us = timeit(lambda: K.affine_act(y, P, Cc, scale, shift, "mish", a1))
    print(f"affine_act (mish)     {us:8.1f} us")


def mfma_probe():
    out = torch.zeros(4, device=DEV)
    for blocks in (1024, 2048, 4096):
        iters = 2000
        us = timeit(lambda: K.mfma_probe(out, blocks, iters), reps=5, warm=2)
        fl = blocks * 4 * 2.0 * iters * 4096
        print(f"mfma_probe blocks={blocks}: {us:8.1f} us  {fl / us / 1e6:7.2f} TFLOP/s (register-only fp32 MFMA)")


if __name__ == "__main__":
    mfma_probe()
    conv_case("3x3 64->64 (RRB conv)", H, W, 64, 64, 3, 3, 1, 1)
    conv_case("3x3 64->64 +bn/mish loader+stats", H, W, 64, 64, 3, 3, 1, 1, prologue=True, stats=True)
    conv_case("1x1 64->64 (GruBlock conv1)", H, W, 64, 64, 1, 1, 0, 0)
    conv_case("1x1 64->192 (GRU input proj)", H, W, 64, 192, 1, 1, 0, 0)
    conv_case("1x1 192->64 (GRU proj dgrad)", H, W, 192, 64, 1, 1, 0, 0)
    conv_case("3x3 64->256 (upsample)", H, W, 64, 256, 3, 3, 1, 1)
    conv_case("9x1 64->36 (tail, HR)", 2 * H, 2 * W, 64, 36, 9, 1, 4, 0)
    conv_case("9x1 36->64 (tail dgrad, HR)", 2 * H, 2 * W, 36, 64, 9, 1, 4, 0)
    conv_case("9x9 4->64 (block1)", H, W, 4, 64, 9, 9, 4, 4)
    gru_case(0)
    gru_case(1)
    bn_case()
