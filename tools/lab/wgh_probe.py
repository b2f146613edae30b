#!/usr/bin/env python
"""Where does the halo weight-gradient kernel's time go?  The recogniser's conv5 (N48 4x26 512->512 3x3) and the up-sampling convolution
(N48 16x64 64->256 3x3) timed with parts of the kernel switched off (tpgsr_wgh_debug: exists in TPGSR_LAB=1 builds only, results are garbage with a bit set)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tpgsr_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
if not hasattr(lib, "tpgsr_wgh_debug"):
    sys.exit("the debug switch of the halo weight-gradient kernel exists only in a lab build: TPGSR_LAB=1 python -m tpgsr_amd.build --force")
lib.tpgsr_wgh_debug.argtypes, lib.tpgsr_wgh_debug.restype = [__import__("ctypes").c_int], __import__("ctypes").c_int
DEV = "cuda"
g = torch.Generator().manual_seed(0)
for (N, H, W, Ci, Co) in ((48, 4, 26, 512, 512), (48, 16, 64, 64, 256), (48, 8, 25, 256, 256)):
    geom = K.ConvGeom(N, H, W, Ci, Co, 3, 3, 1, 1)
    x = torch.randn(geom.M, Ci, generator=g).to(DEV)
    dy = torch.randn(geom.M, Co, generator=g).to(DEV)
    with K.conv_terms(2):
        Z = K.wgrad_splits(geom.M, geom.K, Co, geom=geom)
        part = torch.empty(Z * geom.K * Co, device=DEV)
        dbp = torch.empty(Z * Co, device=DEV)
        w = K.make_wgrad_args(K.make_conv_args(geom, x), dy, part, dbp, zsplits=Z)      # (attaches the dy pre-split scratch)
        for bits, what in ((0, "whole kernel"), (1, "consumers idle (no fragment reads, no MFMAs)"), (2, "producers: no global loads after the first tile"),
                           (4, "producers: no prologue / split / LDS stores"), (6, "producers: barriers only"), (7, "everybody: barriers only")):
            lib.tpgsr_wgh_debug(bits)
            for _ in range(3):
                K.conv_wgrad(w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.conv_wgrad(w)
            e1.record()
            torch.cuda.synchronize()
            print(f"N{N} {H}x{W} {Ci}->{Co} Z={Z}  bits {bits}: {1e3 * e0.elapsed_time(e1) / 20:7.1f} us  ({what}; includes the dy pre-split launch)")
        lib.tpgsr_wgh_debug(0)
