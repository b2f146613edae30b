"""GPU: the whole-CU halo kernel (csrc/conv_halo3.hip: one workgroup per CU on three 64-pixel tiles at once; the SR trunk's 3x3 convolutions,
model/tsrn.py:375-379) against the two-workgroup halo kernel it replaces -- same arithmetic, same summation order per output, so BITWISE the
same outputs and BatchNorm statistics -- and against fp64, over shapes whose 192-pixel super-tiles span rows and images, ragged ends,
every fused prologue, the BatchNorm-backward epilogue, multiple rounds per workgroup and multiple column tiles."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _run(N, H, W, Ci, Co, KH, KW, ph, pw, *, terms, halo3, affine=False, act=False, resid=False, bn=True, bias=True, seed=0, bnb=False):
    from tpgsr_amd import _lib, kernels as K
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N * H * W, Ci, generator=g)
    x2 = torch.randn(N * H * W, Ci, generator=g) if resid else None
    sc = (torch.rand(Ci, generator=g) + 0.5) if affine else None
    sh = (torch.randn(Ci, generator=g) * 0.3) if affine else None
    w = torch.randn(KH * KW * Ci, Co, generator=g) / math.sqrt(Ci * KH * KW)
    b = torch.randn(Co, generator=g) if (bias and not bnb) else None
    geom = K.ConvGeom(N, H, W, Ci, Co, KH, KW, ph, pw)
    dev = lambda t: None if t is None else t.to(DEV).contiguous()
    keep = [dev(x), dev(x2), dev(sc), dev(sh), dev(b)]
    wf = w.to(DEV)
    lib = _lib.load()
    lib.tpgsr_halo3_set_enabled(1 if halo3 else 0)
    try:
        with K.conv_terms(terms):
            K.make_bf_twin(wf, Ci)
            out = torch.full((geom.M, Co), float("nan"), device=DEV)
            part = torch.full(((geom.M + 63) // 64, 2, Co), float("nan"), device=DEV) if (bn or bnb) else None
            kw = {}
            if bnb:      # the data-gradient epilogue: BatchNorm-backward sums behind a mish (tpgsr_conv_args.bnb_*)
                y = torch.randn(geom.M, Co, generator=g).to(DEV)
                mean, rstd = torch.randn(Co, generator=g).to(DEV) * 0.1, (torch.rand(Co, generator=g) + 0.5).to(DEV)
                bsc, bsh = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) * 0.2).to(DEV)
                keep += [y, mean, rstd, bsc, bsh]
                kw = dict(bnb=dict(y=y, mean=mean, rstd=rstd, scale=bsc, shift=bsh, act="mish", partial=part))
            else:
                kw = dict(bn_partial=part)
            K.conv_fwd(K.make_conv_args(geom, keep[0], wf, out, bias=keep[4], in2=keep[1], in_scale=keep[2], in_shift=keep[3],
                                        in_act="mish" if act else None, **kw))
        torch.cuda.synchronize()
    finally:
        lib.tpgsr_halo3_set_enabled(1)
    # fp64 restatement of the forward form
    a = x.double()
    if affine:
        a = a * sc.double() + sh.double()
    if act:
        a = a * torch.tanh(F.softplus(a))
    if resid:
        a = a + x2.double()
    a4 = a.view(N, H, W, Ci).permute(0, 3, 1, 2)
    w4 = w.double().view(KH, KW, Ci, Co).permute(3, 2, 0, 1)
    ref = F.conv2d(a4, w4, b.double() if b is not None else None, padding=(ph, pw)).permute(0, 2, 3, 1).reshape(-1, Co)
    return out, part, ref


SHAPES = [
    # N, H, W, Ci, Co, KH, KW, ph, pw
    (48, 16, 64, 64, 64, 3, 3, 1, 1),      # the trunk: 256 super-tiles = one per CU; every sixth spans two images
    (48, 16, 64, 64, 256, 3, 3, 1, 1),     # upsample conv: four column tiles, four rounds per workgroup
    (13, 16, 64, 64, 64, 3, 3, 1, 1),      # 208 tiles = 69 super-tiles + one tile: ragged last super-tile
    (48, 16, 50, 64, 128, 3, 3, 1, 1),     # recogniser conv1's map: super-tiles start mid-row
    (48, 8, 25, 128, 256, 3, 3, 1, 1),     # 200 pixels per image: nearly every super-tile spans two images; four channel blocks
    (48, 4, 26, 256, 512, 3, 3, 1, 1),     # 104 pixels per image: a super-tile spans up to three images
    (40, 6, 10, 64, 72, 3, 3, 1, 1),       # narrow map (padded width 12), ragged Cout
    (30, 12, 20, 32, 64, 5, 3, 2, 1),      # odd tap count per block (15), one channel block
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("terms", [2, 1])
def test_halo3_bitwise_equals_two_workgroup_halo_kernel(shape, terms):
    from tpgsr_amd import kernels as K
    o3, p3, ref = _run(*shape, terms=terms, halo3=True, seed=11)
    o1, p1, _ = _run(*shape, terms=terms, halo3=False, seed=11)
    err = ((o3.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    print(f"halo3 {shape} T={terms}: err vs fp64 {err:.2e}")
    assert err < {2: 5e-5, 1: 2e-2}[terms]
    assert not torch.isnan(o3).any() and not torch.isnan(p3).any()
    assert torch.equal(o3, o1), f"{(o3 != o1).sum().item()} outputs differ, max {(o3 - o1).abs().max().item():.3e}"
    assert torch.equal(p3, p1), f"{(p3 != p1).sum().item()} statistics differ"


@pytest.mark.parametrize("affine,act,resid", [(True, False, False), (False, True, False), (True, True, False), (False, False, True),
                                               (True, False, True), (True, True, True)])
def test_halo3_prologues(affine, act, resid):
    o3, p3, ref = _run(48, 16, 64, 64, 64, 3, 3, 1, 1, terms=2, halo3=True, affine=affine, act=act, resid=resid, seed=5)
    o1, p1, _ = _run(48, 16, 64, 64, 64, 3, 3, 1, 1, terms=2, halo3=False, affine=affine, act=act, resid=resid, seed=5)
    err = ((o3.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 5e-5
    assert torch.equal(o3, o1) and torch.equal(p3, p1)


def test_halo3_batchnorm_backward_epilogue():
    """the trunk's data gradient with a BatchNorm's backward sums behind a mish in its epilogue (5 of the 11 per step)"""
    o3, p3, _ = _run(48, 16, 64, 64, 64, 3, 3, 1, 1, terms=2, halo3=True, bn=False, bnb=True, seed=7)
    o1, p1, _ = _run(48, 16, 64, 64, 64, 3, 3, 1, 1, terms=2, halo3=False, bn=False, bnb=True, seed=7)
    assert not torch.isnan(p3).any()
    assert torch.equal(o3, o1) and torch.equal(p3, p1)


def test_halo3_is_the_kernel_that_runs_for_the_trunk():
    """guard against a silent fall-through: with the switch on, the trunk shape under x2 must be taken by the whole-CU kernel
    (its launcher reports 1) and under x3 by the two-workgroup kernel (three-term planes do not fit)"""
    import ctypes as C
    from tpgsr_amd import _lib, kernels as K
    lib = _lib.load()
    fn = lib.tpgsr_conv_halo3_xbf_launch
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(_lib.ConvArgs), C.c_longlong, C.c_int, C.c_void_p]
    geom = K.ConvGeom(48, 16, 64, 64, 64, 3, 3, 1, 1)
    x = torch.randn(geom.M, 64, device=DEV)
    wf = torch.randn(576, 64, device=DEV)
    out = torch.empty(geom.M, 64, device=DEV)
    for terms, want in ((2, 1), (3, 0)):
        with K.conv_terms(terms):
            K.make_bf_twin(wf, 64)
            a = K.make_conv_args(geom, x, wf, out)
        assert fn(C.byref(a), geom.M, 0, torch.cuda.current_stream().cuda_stream) == want
    torch.cuda.synchronize()


@pytest.mark.parametrize("terms", [1, 2])
@pytest.mark.parametrize("N,H,W", [(48, 16, 64), (5, 16, 64), (40, 15, 62)])
def test_scaled_residual_loader_is_the_batchnorm_backward_apply(N, H, W, terms):
    """tpgsr_conv_args.in2_scale (round 6): a = in * s + t + in2 * s2 in the loader of the whole-CU kernel == the convolution of the
    MATERIALISED dy = c0 dz + c1 y + c2 (tpgsr_bn_bwd_apply, then the plain loader) -- the folded form of the BatchNorm backward's apply
    (model/tsrn.py:376-380 in the backward pass), with the BatchNorm-backward epilogue of the NEXT BatchNorm riding along as in the trunk.
    Shapes: the trunk's, a ragged batch, and a map whose super-tiles span rows and images."""
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(7)
    C_ = 64
    M = N * H * W
    dz, y = torch.randn(M, C_, generator=g).to(DEV), torch.randn(M, C_, generator=g).to(DEV)
    coef = torch.stack([torch.rand(C_, generator=g) + 0.5, torch.randn(C_, generator=g) * 0.3, torch.randn(C_, generator=g) * 0.1]).to(DEV).contiguous()
    w = (torch.randn(9 * C_, C_, generator=g) / math.sqrt(9 * C_)).to(DEV)
    geom = K.ConvGeom(N, H, W, C_, C_, 3, 3, 1, 1)
    y1 = torch.randn(M, C_, generator=g).to(DEV)
    mean, rstd = (torch.randn(C_, generator=g) * 0.1).to(DEV), (torch.rand(C_, generator=g) + 0.5).to(DEV)
    bsc, bsh = (torch.rand(C_, generator=g) + 0.5).to(DEV), (torch.randn(C_, generator=g) * 0.2).to(DEV)
    with K.conv_terms(terms):
        K.make_bf_twin(w, C_)
        nblk = (M + 63) // 64
        outs = []
        for folded in (True, False):
            out = torch.full((M, C_), float("nan"), device=DEV)
            part = torch.full((nblk, 2, C_), float("nan"), device=DEV)
            bnb = dict(y=y1, mean=mean, rstd=rstd, scale=bsc, shift=bsh, act="mish", partial=part, store_dz=True)
            if folded:
                a = K.make_conv_args(geom, dz, w, out, in_scale=coef[0], in_shift=coef[2], in2=y, in2_scale=coef[1], bnb=bnb)
                ok = K.conv_in2_scale_ok(a)
                if M < 192 * 192:       # fewer super-tiles than the whole-CU kernel asks for: the launch is not its own, and says so
                    assert not ok
                    with pytest.raises(Exception, match="in2_scale"):
                        K.conv_fwd(a)
                    return
                assert ok
                K.conv_fwd(a)
            else:
                dy = torch.empty(M, C_, device=DEV)
                K.bn_bwd_apply(dz, None, y, M, C_, None, None, "none", coef, dy)
                K.conv_fwd(K.make_conv_args(geom, dy, w, out, bnb=bnb))
            outs.append((out, part[:K.bn_rows(M, bnb.get("row_tiles", 1))].clone()))
    torch.cuda.synchronize()
    (o1, p1), (o0, p0) = outs
    scale = o0.abs().max().item()
    e = (o1 - o0).abs().max().item() / scale
    ep = (p1 - p0).abs().max().item() / max(p0.abs().max().item(), 1e-6)
    print(f"folded apply vs materialised dy (terms {terms}, N {N} {H}x{W}): out {e:.2e}, BatchNorm-backward sums {ep:.2e}")
    # the two forms round dy differently (fma chain in the loader vs the apply kernel) before the same split: a few bf16-term ulps
    tol = 2e-2 if terms == 1 else 2e-4
    assert e < tol and ep < 10 * tol and not torch.isnan(o1).any()
