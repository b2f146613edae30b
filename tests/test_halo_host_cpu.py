"""CPU: host-side logic of the halo convolution kernels (csrc/conv_xbf.hip).

* tpgsr_halo_capacity -- the LDS image is sized with it -- must bound the halo length of EVERY 64-pixel tile: checked against a
  brute-force walk over all tiles of random geometries (tiles spanning rows and images included);
* the geometry-only plan of the halo weight-gradient kernel: split count = whole tiles, every split non-empty."""
import ctypes as C
import random

from tpgsr_amd import _lib


def _qbase(m, OH, OW, KH, KW):
    Hp, Wp, ohw = OH + KH - 1, OW + KW - 1, OH * OW
    n, r = divmod(m, ohw)
    oh = r // OW
    return (n * Hp + oh) * Wp + (r - oh * OW)


def test_halo_capacity_bounds_every_tile():
    lib = _lib.load()
    rnd = random.Random(7)
    tight = 0
    for _ in range(4000):
        OH, OW, KH, KW, N = rnd.randint(1, 40), rnd.randint(6, 130), rnd.randint(1, 5), rnd.randint(1, 5), rnd.randint(1, 6)
        if KH * KW < 2:
            continue
        a = _lib.ConvArgs()
        a.OH, a.OW, a.KH, a.KW = OH, OW, KH, KW
        cap = lib.tpgsr_halo_capacity(C.byref(a))
        M, Wp = N * OH * OW, OW + KW - 1
        worst = max(_qbase(min(m0 + 63, M - 1), OH, OW, KH, KW) - _qbase(m0, OH, OW, KH, KW) + (KH - 1) * Wp + KW
                    for m0 in range(0, M, 64))
        assert worst <= cap, (OH, OW, KH, KW, N, worst, cap)
        tight += worst == cap
    assert tight > 0          # the bound is attained, not just safe


def test_wgrad_halo_plan_splits():
    lib = _lib.load()
    for (N, H, W, Ci, Co, Kk) in [(48, 4, 26, 512, 512, 3), (48, 16, 64, 64, 256, 3), (5, 8, 25, 128, 160, 3), (1, 2, 27, 512, 512, 2)]:
        a = _lib.ConvArgs()
        pad = 1 if Kk == 3 else 0
        a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.pad_h, a.pad_w = N, H, W, Ci, Co, Kk, Kk, pad, pad
        a.OH, a.OW, a.terms = H + 2 * pad - Kk + 1, W + 2 * pad - Kk + 1, 3
        z, nb = C.c_int(0), C.c_longlong(0)
        assert lib.tpgsr_wgrad_halo_plan(C.byref(a), C.byref(z), C.byref(nb)) == 1
        M = N * a.OH * a.OW
        tiles = (M + 63) // 64
        tpz = (tiles + z.value - 1) // z.value
        assert 1 <= z.value <= tiles and (z.value - 1) * tpz < tiles            # every split has at least one tile
        assert nb.value == 3 * ((M + 15) // 16) * ((Co + 31) // 32) * 1024
        a.terms = 0
        assert lib.tpgsr_wgrad_halo_plan(C.byref(a), C.byref(z), C.byref(nb)) == 0   # fp32 matrix-core policy: tile loop
    a = _lib.ConvArgs()
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.OH, a.OW, a.terms = 48, 16, 64, 64, 64, 1, 1, 16, 64, 3
    assert lib.tpgsr_wgrad_halo_plan(C.byref(a), None, None) == 0                     # 1x1: not a halo shape
