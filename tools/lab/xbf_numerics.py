#!/usr/bin/env python
"""What v_mfma_f32_32x32x16_bf16 does with small products next to a large accumulator, and the accuracy of the conv kernels
per arithmetic mode against fp64 (rms and max, relative to the rms of the result)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

DEV = "cuda"


def probe(a, b, c, reps=1):
    """a [32][16], b [16][32] fp32 tensors holding bf16-representable values"""
    ab = a.bfloat16().view(torch.int16).contiguous().to(DEV)
    bb = b.bfloat16().view(torch.int16).contiguous().to(DEV)
    cd = c.float().contiguous().to(DEV)
    d = torch.empty(32, 32, device=DEV)
    _lib.check(_lib.load().tpgsr_mfma_bf16_probe(ab.data_ptr(), bb.data_ptr(), cd.data_ptr(), d.data_ptr(), reps,
                                                 torch.cuda.current_stream().cuda_stream), "probe")
    torch.cuda.synchronize()
    return d.cpu()


def main():
    one = torch.ones(32, 32)
    for e in (10, 12, 13, 14):
        a = torch.full((32, 16), 2.0 ** -e)
        b = torch.full((16, 32), 2.0 ** -e)
        d = probe(a, b, one)
        exact = 1.0 + 16 * 2.0 ** (-2 * e)
        print(f"C=1, 16 products of 2^-{2 * e}: D-1 = {float(d[0, 0] - 1):.6e}  exact {exact - 1:.6e}  (fp32 RNE of exact: {float(torch.tensor(exact, dtype=torch.float64).float()) - 1:.6e})")
    # one product of 2^-24 next to C = 1 (half an ulp), 15 zeros
    a = torch.zeros(32, 16); b = torch.zeros(16, 32)
    a[:, 0] = 2.0 ** -12; b[0, :] = 2.0 ** -12
    print("C=1 + single 2^-24 product:", float(probe(a, b, one)[0, 0] - 1))
    a[:, 1] = 2.0 ** -12; b[1, :] = 2.0 ** -12
    print("C=1 + two 2^-24 products (=2^-23 = 1 ulp):", float(probe(a, b, one)[0, 0] - 1))
    a[:, 0] = 1.5 * 2.0 ** -12
    print("C=1 + 1.5*2^-24 + 2^-24 (=2.5*2^-24):", float(probe(a, b, one)[0, 0] - 1))
    # negative tiny products: truncation vs rounding
    a = torch.zeros(32, 16); b = torch.zeros(16, 32)
    a[:, 0] = -1.5 * 2.0 ** -12; b[0, :] = 2.0 ** -12
    print("C=1 - 1.5*2^-24:", float(probe(a, b, one)[0, 0] - 1), " (RNE: -1.19e-07 = -2^-23; toward zero: -5.96e-08)")
    # random accumulation drift: repeat the same MFMA 256 times with small random products into a big accumulator
    torch.manual_seed(0)
    a = (torch.randn(32, 16) * 2.0 ** -9).bfloat16().float(); b = (torch.randn(16, 32)).bfloat16().float()
    c0 = torch.randn(32, 32) * 4
    for reps in (1, 64, 1024):
        d = probe(a, b, c0, reps)
        exact = c0.double() + reps * (a.double() @ b.double())
        # fp32 emulation: c <- fl(c + exact_block)
        emu = c0.clone()
        blk = (a.double() @ b.double())
        for _ in range(reps):
            emu = (emu.double() + blk).float()
        print(f"reps {reps}: hw vs exact rel {float((d.double() - exact).abs().max() / exact.abs().max()):.3e}; RNE-per-MFMA emulation vs exact "
              f"{float((emu.double() - exact).abs().max() / exact.abs().max()):.3e}; mean signed err hw {float((d.double() - exact).mean()):.3e} emu {float((emu.double() - exact).mean()):.3e}")

    # conv accuracy per mode
    print("\n| shape | mode | rms err / rms ref | max err / rms ref |")
    print("|---|---|---|---|")
    g = torch.Generator().manual_seed(1)
    for (N, H, W, Ci, Co, KH, pad) in [(4, 16, 64, 64, 64, 3, 1), (8, 4, 26, 512, 512, 3, 1), (48, 1, 26, 512, 2048, 1, 0), (2, 32, 128, 64, 36, 9, 4)]:
        KW = KH if KH != 9 else 1
        pw = pad if KH != 9 else 0
        x = torch.randn(N, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, KH, KW, generator=g) / math.sqrt(Ci * KH * KW)
        ref = F.conv2d(x.double(), w.double(), None, padding=(pad, pw))
        geom = K.ConvGeom(N, H, W, Ci, Co, KH, KW, pad, pw)
        wf = w.permute(2, 3, 1, 0).reshape(-1, Co).contiguous().to(DEV)
        K.make_bf_twin(wf)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        for mode in ("f32", "x3", "bf16"):
            K.set_conv_prec(mode)
            out = torch.empty(geom.M, Co, device=DEV)
            K.conv_fwd(K.make_conv_args(geom, xd, wf, out))
            torch.cuda.synchronize()
            got = out.reshape(N, geom.OH, geom.OW, Co).permute(0, 3, 1, 2).cpu().double()
            e = got - ref
            rms = ref.pow(2).mean().sqrt()
            print(f"| {N}x{H}x{W} {Ci}->{Co} k{KH}x{KW} (K={geom.K}) | {mode} | {float(e.pow(2).mean().sqrt() / rms):.3e} | {float(e.abs().max() / rms):.3e} |")
        K.set_conv_prec("f32")


if __name__ == "__main__":
    main()
