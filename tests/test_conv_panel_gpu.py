"""GPU: the row-panel kernel of the 1x1 convolutions (csrc/conv_panel.hip: 64 pixels x all output columns per workgroup, the
whole K at once) against fp64, on the GruBlock projection shapes of the SR network (model/tsrn.py:491-508) with every loader it is
instantiated for, ragged pixel counts, BN statistics, column-block weight views, and in all three split arithmetics."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def small_panels():
    """let the panel kernel take small pixel counts too (default threshold: 32768 pixels)"""
    from tpgsr_amd import _lib
    lib = _lib.load()
    lib.tpgsr_panel_set_min_m(64)
    lib.tpgsr_panel_set_k192(1)
    yield lib
    lib.tpgsr_panel_set_min_m(32768)
    lib.tpgsr_panel_set_k192(0)


CASES = [
    # (N, H, W, Ci, Co), loader / epilogue options
    ((48, 16, 64, 192, 64), dict(affine=False, act=False, resid=False, bn=False, bias=False)),    # projection data gradient (full size)
    ((48, 16, 64, 64, 192), dict(affine=False, act=False, resid=True, bn=False, bias=True)),      # gru2: residual-add loader (LD 4)
    ((48, 16, 64, 64, 192), dict(affine=True, act=False, resid=False, bn=False, bias=True)),      # gru1 without text prior (LD 1)
    ((3, 5, 9, 64, 192), dict(affine=True, act=False, resid=False, bn=True, bias=True)),           # ragged M (135 pixels), BN statistics
    ((3, 5, 9, 192, 40), dict(affine=False, act=False, resid=False, bn=True, bias=True)),          # ragged Cout
    ((2, 7, 13, 96, 192), dict(affine=False, act=False, resid=False, bn=False, bias=True)),        # K = 96
    ((2, 7, 13, 64, 100), dict(affine=False, act=False, resid=True, bn=True, bias=False)),         # four column blocks, last one partial
]


@pytest.mark.parametrize("shape,kw", CASES)
def test_panel_kernel_vs_fp64(shape, kw, small_panels):
    import test_conv_xbf_gpu as X
    N, H, W, Ci, Co = shape
    e_out, e_bn = X._halo_case(N, H, W, Ci, Co, 1, 1, 0, 0, seed=31, **kw)
    print(f"panel {shape} {kw}: out {e_out:.2e}  bn {e_bn:.2e}")
    assert e_out < 3e-6 and e_bn < 2e-5
    # and the tile loop on the same case (the switch really routes)
    small_panels.tpgsr_panel_set_enabled(0)
    try:
        e2, _ = X._halo_case(N, H, W, Ci, Co, 1, 1, 0, 0, seed=31, **kw)
    finally:
        small_panels.tpgsr_panel_set_enabled(1)
    assert e2 < 3e-6


@pytest.mark.parametrize("terms,tol", [(3, 3e-6), (2, 4e-5), (1, 2e-2)])
def test_panel_concat_loader_and_weight_column_blocks(terms, tol, small_panels):
    """gru1 of TSRN_TL: [BN-affine(y2) || text strip broadcast over H] -> 192 (LD 17), and the data gradient of that projection in two
    column blocks of ONE packed operand (wt_ld = 96: columns 0..63 image features, 64..95 text strip), as engine.py records them"""
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(77)
    N, H, W, Ca, Cb, Co = 5, 4, 11, 64, 32, 192
    M = N * H * W
    xa = torch.randn(M, Ca, generator=g)
    strip = torch.randn(N * W, Cb, generator=g)
    sc, sh = torch.rand(Ca + Cb, generator=g) + 0.5, torch.randn(Ca + Cb, generator=g) * 0.3
    sc[Ca:], sh[Ca:] = 1.0, 0.0                                     # identity tail (BNLayer pad_to)
    w = torch.randn(Ca + Cb, Co, generator=g) / math.sqrt(Ca + Cb)  # [K][Cout]
    bias = torch.randn(Co, generator=g)
    a = xa.double() * sc[:Ca].double() + sh[:Ca].double()
    full = torch.cat([a.view(N, H, W, Ca), strip.double().view(N, 1, W, Cb).expand(N, H, W, Cb)], -1).reshape(M, Ca + Cb)
    ref = full @ w.double() + bias.double()
    with K.conv_terms(terms):
        wf = w.contiguous().to(DEV)
        K.make_bf_twin(wf)
        out = torch.full((M, Co), float("nan"), device=DEV)
        keep = [xa.to(DEV), strip.to(DEV), sc.to(DEV), sh.to(DEV), bias.to(DEV)]
        K.conv_fwd(K.make_conv_args(K.ConvGeom(N, H, W, Ca + Cb, Co), keep[0], wf, out, bias=keep[4], in_b=keep[1], cin_a=Ca,
                                    in_scale=keep[2], in_shift=keep[3]))
        # data gradient: dgi [M][192] x wd [192][96], column blocks 0..63 and 64..95
        dgi = torch.randn(M, Co, generator=g)
        wd = torch.randn(Co, Ca + Cb, generator=g) / math.sqrt(Co)
        wdd = wd.contiguous().to(DEV)
        K.make_bf_twin(wdd)
        dgid = dgi.to(DEV)
        da, dtb = torch.full((M, Ca), float("nan"), device=DEV), torch.full((M, Cb), float("nan"), device=DEV)
        K.conv_fwd(K.make_conv_args(K.ConvGeom(N, H, W, Co, Ca), dgid, wdd, da, wt_ld=Ca + Cb, wt_coff=0))
        K.conv_fwd(K.make_conv_args(K.ConvGeom(N, H, W, Co, Cb), dgid, wdd, dtb, wt_ld=Ca + Cb, wt_coff=Ca))
    torch.cuda.synchronize()
    rd = dgi.double() @ wd.double()
    errs = [((out.cpu().double() - ref).abs().max() / ref.abs().max()).item(),
            ((da.cpu().double() - rd[:, :Ca]).abs().max() / rd.abs().max()).item(),
            ((dtb.cpu().double() - rd[:, Ca:]).abs().max() / rd.abs().max()).item()]
    print(f"panel concat / column blocks, terms {terms}: {errs}")
    assert max(errs) < tol
