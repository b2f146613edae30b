"""GPU: the tile-loop weight gradient with three k-blocks per workgroup (csrc/conv_xbf.hip: conv_wgrad_xbf3_kernel, round 6) against the
one-block kernel it replaces (same slabs, same summation order: BITWISE with the plain loader) and against fp64 autograd -- the trunk's
3x3 64 -> 64 at full batch (model/tsrn.py:375-379), shapes whose 32-pixel chunks straddle rows and images, narrow maps (several row wraps per
chunk), ragged splits, Cout that is no multiple of 64, the affine loader, the bias-gradient partials."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(N, H, W, Ci, Co, KH, KW, terms, affine, three, Z=0, seed=0):
    from tpgsr_amd import _lib, kernels as K
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N * H * W, Ci, generator=g).to(DEV)
    dy = torch.randn(N * H * W, Co, generator=g).to(DEV)
    sc = (torch.rand(Ci, generator=g) + 0.5).to(DEV) if affine else None
    sh = (torch.randn(Ci, generator=g) * 0.3).to(DEV) if affine else None
    geom = K.ConvGeom(N, H, W, Ci, Co, KH, KW, KH // 2, KW // 2)
    lib = _lib.load()
    lib.tpgsr_wgrad3_set_enabled(1 if three else 0)
    try:
        with K.conv_terms(terms):
            Zs = Z or lib.tpgsr_wgrad_splits(geom.M, geom.K, Co)
            part = torch.full((Zs, geom.K, Co), float("nan"), device=DEV)
            dbp = torch.full((Zs, Co), float("nan"), device=DEV)
            wa = K.make_wgrad_args(K.make_conv_args(geom, x, in_scale=sc, in_shift=sh), dy, part, dbp)
            wa.zsplits = Z                      # (set directly: make_wgrad_args(zsplits=) would also ask the halo kernel, which 3x3 shapes get)
            wa.dy_bf = None
            K.conv_wgrad(wa)
        torch.cuda.synchronize()
    finally:
        lib.tpgsr_wgrad3_set_enabled(1)
    return part, dbp, (x, dy, sc, sh, geom)


def _fp64(x, dy, sc, sh, geom):
    N, H, W, Ci, Co = geom.N, geom.H, geom.W, geom.Cin, geom.Cout
    a = x.double().cpu()
    if sc is not None:
        a = a * sc.double().cpu() + sh.double().cpu()
    a4 = a.view(N, H, W, Ci).permute(0, 3, 1, 2)
    w = torch.zeros(Co, Ci, geom.KH, geom.KW, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(a4, w, padding=(geom.pad_h, geom.pad_w))
    y.backward(dy.double().cpu().view(N, H, W, Co).permute(0, 3, 1, 2))
    # part layout: [k = (kh KW + kw) Ci + ci][co]
    return w.grad.permute(2, 3, 1, 0).reshape(-1, Co), dy.double().cpu().sum(0)


@pytest.mark.parametrize("terms", [1, 2])
@pytest.mark.parametrize("shape", [(48, 16, 64, 64, 64, 3, 3), (3, 8, 25, 64, 40, 3, 3), (2, 5, 7, 64, 128, 3, 3), (4, 4, 26, 64, 100, 1, 3),
                                   (2, 16, 64, 192, 64, 1, 1)])
@pytest.mark.parametrize("affine", [False, True])
def test_three_k_blocks_equal_one(shape, affine, terms):
    # (all shapes have Cin x Cout < 16384: the halo weight-gradient kernel leaves them to the tile loop)
    N, H, W, Ci, Co, KH, KW = shape
    p3, b3, ops = _run(N, H, W, Ci, Co, KH, KW, terms, affine, True)
    p1, b1, _ = _run(N, H, W, Ci, Co, KH, KW, terms, affine, False)
    assert not torch.isnan(p3).any() and not torch.isnan(b3).any()
    if not affine:
        assert torch.equal(p3, p1) and torch.equal(b3, b1)          # same MFMA sequence per output element
    ref_w, ref_b = _fp64(*ops)
    dw = p3.double().sum(0).cpu()
    e = (dw - ref_w).abs().max().item() / ref_w.abs().max().item()
    e1 = (p1.double().sum(0).cpu() - ref_w).abs().max().item() / ref_w.abs().max().item()
    eb = (b3.double().sum(0).cpu() - ref_b).abs().max().item() / ref_b.abs().max().item()
    print(f"wgrad3 {shape} affine {affine} x{terms}: dW rel err {e:.2e} (one-block kernel {e1:.2e}), db {eb:.2e}")
    tol = 2e-2 if terms == 1 else 5e-5
    assert e < tol and e <= 2 * e1 + 1e-7 and eb < 1e-5


def test_three_k_blocks_with_an_explicit_split_count():
    for Z in (1, 5, 37):
        p3, b3, ops = _run(6, 16, 64, 64, 64, 3, 3, 2, False, True, Z=Z, seed=Z)
        p1, b1, _ = _run(6, 16, 64, 64, 64, 3, 3, 2, False, False, Z=Z, seed=Z)
        assert torch.equal(p3, p1) and torch.equal(b3, b1), Z
