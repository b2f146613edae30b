#!/usr/bin/env python
"""Per-shape timing of the MFMA conv kernels in the three arithmetic modes (f32 matrix cores / x3 = fp32-equivalent split on
the bf16 matrix cores / bf16 operands), forward and weight gradient, at the C3 shapes (bs 48).  HIP-event timing on the
launch stream, interleaved rounds (one process, median of rounds).  Prints a markdown table."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tpgsr_amd import kernels as K  # noqa: E402

DEV = "cuda"


def time_fns(fns, rounds=7, reps=8):
    res = [[] for _ in fns]
    for fn in fns:
        fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for i, fn in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[i].append(1e3 * e0.elapsed_time(e1) / reps)
    return [statistics.median(r) for r in res]


def case(name, N, Hh, Ww, Ci, Co, KH, KW, ph, pw):
    g = K.ConvGeom(N, Hh, Ww, Ci, Co, KH, KW, ph, pw)
    x = torch.randn(g.N * Hh * Ww, Ci, device=DEV)
    wf = torch.randn(g.K, Co, device=DEV) * 0.05
    out = torch.empty(g.M, Co, device=DEV)
    b = torch.randn(Co, device=DEV)
    dy = torch.randn(g.M, Co, device=DEV)
    K.make_bf_twin(wf, Ci)
    fa, wa, keep = [], [], []
    for prec in ("f32", "x3", "bf16"):
        K.set_conv_prec(prec)
        fa.append(K.make_conv_args(g, x, wf, out, bias=b))
        Z = K.wgrad_splits(g.M, g.K, Co, geom=g)          # (the halo weight-gradient kernel's own split count where it applies)
        part = torch.empty(Z, g.K, Co, device=DEV)
        keep.append(part)
        wa.append(K.make_wgrad_args(K.make_conv_args(g, x), dy, part, None, zsplits=Z))
    K.set_conv_prec("f32")
    tf = time_fns([lambda a=a: K.conv_fwd(a) for a in fa])
    tw = time_fns([lambda a=a: K.conv_wgrad(a) for a in wa])
    fl = 2.0 * g.M * g.K * Co
    def tf_(us):
        return f"{us:7.1f} us {fl / us / 1e6:6.1f} TF"
    print(f"| {name} | {fl / 1e9:.2f} | {tf_(tf[0])} | {tf_(tf[1])} | {tf_(tf[2])} | {tf_(tw[0])} | {tf_(tw[1])} | {tf_(tw[2])} |", flush=True)


if __name__ == "__main__":
    print("| shape (bs 48) | GFLOP | fwd f32 | fwd x3 | fwd bf16 | wgrad f32 | wgrad x3 | wgrad bf16 |")
    print("|---|---|---|---|---|---|---|---|")
    B = 48
    case("TSRN 3x3 64->64 @16x64", B, 16, 64, 64, 64, 3, 3, 1, 1)
    case("TSRN 1x1 64->192 (GRU proj)", B, 16, 64, 64, 192, 1, 1, 0, 0)
    case("TSRN 1x1 192->64 (GRU dgrad)", B, 16, 64, 192, 64, 1, 1, 0, 0)
    case("TSRN 3x3 64->256 (upsample)", B, 16, 64, 64, 256, 3, 3, 1, 1)
    case("TSRN 9x1 64->36 @32x128 (tail)", B, 32, 128, 64, 36, 9, 1, 4, 0)
    case("TSRN 9x9 4->64 (block1)", B, 16, 64, 4, 64, 9, 9, 4, 4)
    case("CRNN conv1 3x3 64->128 @16x50", B, 16, 50, 64, 128, 3, 3, 1, 1)
    case("CRNN conv2 3x3 128->256 @8x25", B, 8, 25, 128, 256, 3, 3, 1, 1)
    case("CRNN conv3 3x3 256->256 @8x25", B, 8, 25, 256, 256, 3, 3, 1, 1)
    case("CRNN conv4 3x3 256->512 @4x26", B, 4, 26, 256, 512, 3, 3, 1, 1)
    case("CRNN conv5 3x3 512->512 @4x26", B, 4, 26, 512, 512, 3, 3, 1, 1)
    case("CRNN conv6 2x2 512->512 @2x27", B, 2, 27, 512, 512, 2, 2, 0, 0)
    case("CRNN LSTM in-proj 512->2048 (T=26)", B, 1, 26, 512, 2048, 1, 1, 0, 0)
