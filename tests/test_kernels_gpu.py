"""GPU: every C-ABI kernel against a plain PyTorch-CPU (fp64 where cheap) restatement of the same op.
Tolerances are written next to each check; fp32 MFMA == fmaf chain, so conv errors are accumulation-order only."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tpgsr_oracle as O  # noqa: E402


def K():
    from tpgsr_amd import kernels
    return kernels


DEV = "cuda"


def to_nhwc(x):  # NCHW cpu -> NHWC cuda contiguous
    return x.permute(0, 2, 3, 1).contiguous().float().to(DEV)


def from_nhwc(t, N, H, W, C):  # NHWC cuda flat -> NCHW cpu
    return t.reshape(N, H, W, C).permute(0, 3, 1, 2).cpu()


def relerr(got, ref):
    got, ref = got.double(), ref.double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def pack_f(w):  # [Co][Ci][KH][KW] -> [K][Co] on device through the kernel under test
    Co, Ci, KH, KW = w.shape
    k = K()
    wf = torch.empty(KH * KW * Ci, Co, device=DEV)
    wd = torch.empty(KH * KW * Co, Ci, device=DEV)
    wdev = w.to(DEV).contiguous()
    k.pack_conv_weight(wdev, Co, Ci, KH, KW, wf, wd)
    if k.CONV_TERMS:
        k.make_bf_twin(wf, Ci)
        k.make_bf_twin(wd, Co)
    torch.cuda.synchronize()
    return wf, wd


@pytest.fixture(params=["f32", "x3"])
def prec(request):
    """MFMA arithmetic of the conv / linear GEMMs: fp32 matrix cores, or the fp32-equivalent split-operand path on the
    bf16 matrix cores (csrc/conv_xbf.hip) -- the SAME tolerances must hold for both."""
    k = K()
    prev = k.POLICY
    k.set_conv_prec(request.param)
    yield request.param
    k.set_conv_prec(prev)


def mish(x):
    return x * torch.tanh(F.softplus(x))


CONV_CASES = [
    # N, H, W, Cin, Cout, KH, KW, ph, pw
    (2, 16, 64, 64, 64, 3, 3, 1, 1),
    (2, 16, 64, 4, 64, 9, 9, 4, 4),
    (3, 7, 13, 64, 96, 1, 1, 0, 0),      # ragged M, Cout not a multiple of 64
    (2, 8, 24, 32, 64, 3, 3, 1, 1),
    (2, 6, 10, 1, 64, 3, 3, 1, 1),       # scalar loader (CRNN conv0)
    (2, 5, 9, 3, 32, 5, 5, 2, 2),        # scalar loader, Cin = 3
    (5, 1, 1, 512, 40, 1, 1, 0, 0),      # linear layer (STN fc2)
    (4, 1, 2, 256, 512, 1, 2, 0, 0),     # STN fc1 as a 1x2 valid conv
    (2, 2, 27, 512, 512, 2, 2, 0, 0),    # CRNN conv6 (valid 2x2)
    (3, 1, 1, 64, 37, 1, 1, 0, 0),       # Cout = 37: scalar weight loads
    (1, 32, 128, 64, 36, 9, 1, 4, 0),    # folded tail conv
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_plain(case, prec):
    N, H, W, Ci, Co, KH, KW, ph, pw = case
    k = K()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, KH, KW, generator=g) / math.sqrt(Ci * KH * KW)
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=(ph, pw))
    geom = k.ConvGeom(N, H, W, Ci, Co, KH, KW, ph, pw)
    wf, wd = pack_f(w)
    out = torch.full((geom.M, Co), float("nan"), device=DEV)
    xd, bd = to_nhwc(x), b.to(DEV)
    k.conv_fwd(k.make_conv_args(geom, xd, wf, out, bias=bd))
    torch.cuda.synchronize()
    got = from_nhwc(out, N, geom.OH, geom.OW, Co)
    assert relerr(got, ref) < 5e-6, relerr(got, ref)   # fp32 accumulation-order noise only
    # data gradient == conv over dy with the dgrad packing
    dy = torch.randn(ref.shape, generator=g)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, padding=(ph, pw)).backward(dy.double())
    gd = geom.dgrad()
    dx = torch.full((N * H * W, Ci), float("nan"), device=DEV)
    dyd = to_nhwc(dy)
    k.conv_fwd(k.make_conv_args(gd, dyd, wd, dx))
    torch.cuda.synchronize()
    assert relerr(from_nhwc(dx, N, H, W, Ci), xr.grad) < 5e-6


@pytest.mark.parametrize("Cout,ks,ps", [(192, 1, False), (256, 3, True), (192, 3, False)])
def test_conv_fwd_wide_tiles(Cout, ks, ps, prec):
    """Wide outputs (Cout = 192 / 256: several column tiles per pixel tile) on a grid of > 512 pixel tiles with a ragged
    last one: loader affine + residual, bias, optional pixel-shuffle store."""
    k = K()
    N, H, W, C = 9, 57, 64, 64                       # M = 32832 = 513 tiles
    g = torch.Generator().manual_seed(Cout + ks)
    x = torch.randn(N, C, H, W, generator=g)
    x2 = torch.randn(N, C, H, W, generator=g)
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    w = torch.randn(Cout, C, ks, ks, generator=g) / math.sqrt(C * ks * ks)
    b = torch.randn(Cout, generator=g)
    a = x.double() * sc.view(1, -1, 1, 1).double() + sh.view(1, -1, 1, 1).double() + x2.double()
    ref = F.conv2d(a, w.double(), b.double(), padding=ks // 2)
    geom = k.ConvGeom(N, H, W, C, Cout, ks, ks, ks // 2, ks // 2)
    assert (geom.M + 63) // 64 >= 512
    wf, _ = pack_f(w)
    keep = [to_nhwc(x), b.to(DEV), to_nhwc(x2), sc.to(DEV), sh.to(DEV)]
    out = torch.full((geom.M * (4 if ps else 1), Cout // (4 if ps else 1)), float("nan"), device=DEV)
    k.conv_fwd(k.make_conv_args(geom, keep[0], wf, out, bias=keep[1], in2=keep[2], in_scale=keep[3], in_shift=keep[4], out_ps=ps))
    torch.cuda.synchronize()
    if ps:
        got, want = from_nhwc(out, N, 2 * H, 2 * W, Cout // 4), F.pixel_shuffle(ref, 2)
    else:
        got, want = from_nhwc(out, N, H, W, Cout), ref
    assert not torch.isnan(out).any()
    assert relerr(got, want) < 3e-6


def test_conv_fwd_prologue_epilogue_bnstats(prec):
    """loader: mish(scale*x+shift) + in2 ; epilogue: bias, relu, per-block BN partial sums."""
    k = K()
    N, H, W, C = 2, 16, 64, 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, C, H, W, generator=g)
    x2 = torch.randn(N, C, H, W, generator=g)
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b = torch.randn(C, generator=g)
    a = mish(x.double() * sc.view(1, -1, 1, 1).double() + sh.view(1, -1, 1, 1).double()) + x2.double()
    pre = F.conv2d(a, w.double(), b.double(), padding=1)
    geom = k.ConvGeom(N, H, W, C, C, 3, 3, 1, 1)
    wf, _ = pack_f(w)
    out = torch.empty(geom.M, C, device=DEV)
    nblk = (geom.M + 63) // 64
    part = torch.zeros(nblk, 2, C, device=DEV)
    keep = [to_nhwc(x), b.to(DEV), to_nhwc(x2), sc.to(DEV), sh.to(DEV)]
    k.conv_fwd(k.make_conv_args(geom, keep[0], wf, out, bias=keep[1], in2=keep[2], in_scale=keep[3],
                                in_shift=keep[4], in_act="mish", out_act="relu", bn_partial=part))
    torch.cuda.synchronize()
    assert relerr(from_nhwc(out, N, H, W, C), F.relu(pre)) < 3e-6
    raw = (pre - b.double().view(1, -1, 1, 1))          # partials are statistics of (out - bias), pre-activation
    s = part[:, 0].double().sum(0).cpu()
    ss = part[:, 1].double().sum(0).cpu()
    assert relerr(s, raw.sum((0, 2, 3))) < 1e-5
    assert relerr(ss, (raw ** 2).sum((0, 2, 3))) < 1e-5
    # finalize -> scale/shift/save stats/running stats like nn.BatchNorm2d(train)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    rm = torch.randn(C, generator=g) * 0.1
    rv = torch.rand(C, generator=g) + 0.5
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    scale = torch.empty(C, device=DEV); shift = torch.empty(C, device=DEV)
    smean = torch.empty(C, device=DEV); srstd = torch.empty(C, device=DEV)
    gd_, bt_ = gamma.to(DEV), beta.to(DEV)
    k.bn_finalize(part, nblk, C, geom.M, keep[1], gd_, bt_, rm_d, rv_d, scale, shift, smean, srstd)
    torch.cuda.synchronize()
    rm_ref, rv_ref = rm.clone().double(), rv.clone().double()
    y_ref = F.batch_norm(pre, rm_ref, rv_ref, gamma.double(), beta.double(), True, 0.1, 1e-5)
    y_got = pre * scale.cpu().double().view(1, -1, 1, 1) + shift.cpu().double().view(1, -1, 1, 1)
    assert (y_got - y_ref).abs().max() < 2e-5
    assert (rm_d.cpu().double() - rm_ref).abs().max() < 1e-6 and (rv_d.cpu().double() - rv_ref).abs().max() < 1e-6
    # eval mode: scale/shift from running stats
    k.bn_finalize(None, 0, C, 0, None, gd_, bt_, rm_d, rv_d, scale, shift, eval_mode=True)
    torch.cuda.synchronize()
    y_ref = F.batch_norm(pre, rm_ref, rv_ref, gamma.double(), beta.double(), False, 0.1, 1e-5)
    y_got = pre * scale.cpu().double().view(1, -1, 1, 1) + shift.cpu().double().view(1, -1, 1, 1)
    assert (y_got - y_ref).abs().max() < 2e-5


def test_conv_pixel_shuffle_store_and_gather(prec):
    """out_ps writes nn.PixelShuffle(2) layout; in_ps / dy_ps read it back as the un-shuffled tensor."""
    k = K()
    N, H, W, C = 2, 8, 24, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(4 * C, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b = torch.randn(4 * C, generator=g)
    pre = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.pixel_shuffle(pre, 2)
    geom = k.ConvGeom(N, H, W, C, 4 * C, 3, 3, 1, 1)
    wf, wd = pack_f(w)
    out = torch.empty(N * 2 * H * 2 * W, C, device=DEV)
    xd, bd = to_nhwc(x), b.to(DEV)
    k.conv_fwd(k.make_conv_args(geom, xd, wf, out, bias=bd, out_ps=True))
    torch.cuda.synchronize()
    assert relerr(from_nhwc(out, N, 2 * H, 2 * W, C), ref) < 2e-6
    # dgrad through the shuffle: dy given in shuffled layout
    dyps = torch.randn(ref.shape, generator=g)
    xr = x.double().requires_grad_(True)
    F.pixel_shuffle(F.conv2d(xr, w.double(), None, padding=1), 2).backward(dyps.double())
    gd = geom.dgrad()
    dx = torch.empty(N * H * W, C, device=DEV)
    dyd = to_nhwc(dyps)
    k.conv_fwd(k.make_conv_args(gd, dyd, wd, dx, in_ps=True))
    torch.cuda.synchronize()
    assert relerr(from_nhwc(dx, N, H, W, C), xr.grad) < 2e-6
    # wgrad with dy_ps
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    F.pixel_shuffle(F.conv2d(x.double(), wr, br, padding=1), 2).backward(dyps.double())
    Z = k.wgrad_splits(geom.M, geom.K, geom.Cout)
    part = torch.empty(Z, geom.K, geom.Cout, device=DEV)
    dbp = torch.empty(Z, geom.Cout, device=DEV)
    ca = k.make_conv_args(geom, xd)
    k.conv_wgrad(k.make_wgrad_args(ca, dyd, part, dbp, dy_ps=True))
    dw = torch.zeros_like(w, device=DEV); db = torch.zeros(4 * C, device=DEV)
    k.wgrad_reduce(part, dbp, Z, geom, dw, db, accumulate=False)
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), wr.grad) < 5e-6 and relerr(db.cpu(), br.grad) < 5e-6


@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[1], CONV_CASES[2], CONV_CASES[4], CONV_CASES[6], CONV_CASES[9]])
def test_conv_wgrad(case, prec):
    N, H, W, Ci, Co, KH, KW, ph, pw = case
    k = K()
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = (torch.randn(Co, Ci, KH, KW, generator=g)).double().requires_grad_(True)
    b = torch.randn(Co, generator=g).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, b, padding=(ph, pw))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    geom = k.ConvGeom(N, H, W, Ci, Co, KH, KW, ph, pw)
    Z = k.wgrad_splits(geom.M, geom.K, Co)
    part = torch.full((Z, geom.K, Co), float("nan"), device=DEV)
    dbp = torch.full((Z, Co), float("nan"), device=DEV)
    xd, dyd = to_nhwc(x), to_nhwc(dy)
    k.conv_wgrad(k.make_wgrad_args(k.make_conv_args(geom, xd), dyd, part, dbp))
    dw = torch.ones(Co, Ci, KH, KW, device=DEV); db = torch.ones(Co, device=DEV)
    k.wgrad_reduce(part, dbp, Z, geom, dw, db, accumulate=True)       # += on top of ones
    torch.cuda.synchronize()
    assert relerr(dw.cpu() - 1, w.grad) < 1e-5, relerr(dw.cpu() - 1, w.grad)
    assert relerr(db.cpu() - 1, b.grad) < 1e-5


def test_wgrad_with_loader_prologue(prec):
    k = K()
    N, H, W, C = 2, 16, 64, 64
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, C, H, W, generator=g); x2 = torch.randn(N, C, H, W, generator=g)
    sc = torch.rand(C, generator=g) + 0.5; sh = torch.randn(C, generator=g) * 0.3
    a = mish(x.double() * sc.view(1, -1, 1, 1).double() + sh.view(1, -1, 1, 1).double()) + x2.double()
    w = torch.randn(C, C, 3, 3, generator=g).double().requires_grad_(True)
    y = F.conv2d(a, w, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    geom = k.ConvGeom(N, H, W, C, C, 3, 3, 1, 1)
    Z = k.wgrad_splits(geom.M, geom.K, C)
    part = torch.empty(Z, geom.K, C, device=DEV)
    keep = [to_nhwc(x), to_nhwc(x2), sc.to(DEV), sh.to(DEV), to_nhwc(dy)]
    ca = k.make_conv_args(geom, keep[0], in2=keep[1], in_scale=keep[2], in_shift=keep[3], in_act="mish")
    k.conv_wgrad(k.make_wgrad_args(ca, keep[4], part))
    dw = torch.zeros(C, C, 3, 3, device=DEV)
    k.wgrad_reduce(part, None, Z, geom, dw, None, accumulate=False)
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), w.grad) < 1e-5


def test_tail_fold_matches_conv9x9(prec):
    """9x9 C->4 conv + tanh == 9x1 conv with 36 folded columns + shift-sum; and its backward."""
    k = K()
    N, H, W, C, Co, KS = 2, 12, 40, 64, 4, 9
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, C, H, W, generator=g)
    w = (torch.randn(Co, C, KS, KS, generator=g) / math.sqrt(C * 81)).double().requires_grad_(True)
    b = torch.randn(Co, generator=g).double().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    ref = torch.tanh(F.conv2d(mish(xr), w, b, padding=4))
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout.double())
    wf = torch.empty(KS * C, KS * Co, device=DEV); wd = torch.empty(KS * KS * Co, C, device=DEV)
    wdev = w.detach().float().to(DEV).contiguous()
    k.pack_tail_weight(wdev, Co, C, KS, wf, wd)
    if k.CONV_TERMS:
        k.make_bf_twin(wf, C)
        k.make_bf_twin(wd)
    geom = k.ConvGeom(N, H, W, C, KS * Co, KS, 1, KS // 2, 0)
    P = torch.empty(geom.M, KS * Co, device=DEV)
    xd = to_nhwc(x)
    ca = k.make_conv_args(geom, xd, wf, P, in_act="mish")
    k.conv_fwd(ca)
    out = torch.empty(N, Co, H, W, device=DEV)
    bd = b.detach().float().to(DEV)
    k.tail_shiftsum_tanh(P, bd, N, H, W, Co, KS, out)
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref.detach()) < 3e-6
    nblk = k.tail_bwd_blocks(N, H, W, Co, KS)
    dP = torch.empty_like(P); dbp = torch.empty(nblk, Co, device=DEV)
    doutd = dout.to(DEV)
    k.tail_bwd(out, doutd, N, H, W, Co, KS, dP, dbp, nblk)
    db = torch.zeros(Co, device=DEV)
    k.reduce_partials(dbp, nblk, Co, db, accumulate=False)
    Z = k.wgrad_splits(geom.M, geom.K, geom.Cout)
    part = torch.empty(Z, geom.K, geom.Cout, device=DEV)
    k.conv_wgrad(k.make_wgrad_args(ca, dP, part))
    dw = torch.zeros(Co, C, KS, KS, device=DEV)
    k.wgrad_reduce(part, None, Z, geom, dw, None, layout=2, accumulate=False)
    dm = torch.empty(N * H * W, C, device=DEV)
    k.conv_fwd(k.make_conv_args(geom.dgrad(), dP, wd, dm))
    dx = torch.empty_like(dm)
    k.act_bwd(xd, dm, dm.numel(), "mish", dx)
    torch.cuda.synchronize()
    assert relerr(db.cpu(), b.grad) < 1e-5
    assert relerr(dw.cpu(), w.grad) < 1e-5
    assert relerr(from_nhwc(dx, N, H, W, C), xr.grad) < 1e-5


@pytest.mark.parametrize("W", [128, 200, 333])
def test_tail_shiftsum_and_backward_rows_wider_than_one_workgroup(W):
    """rows wider than the LDS block (W (KS Co + 1) > 5920 floats forward, W > 160 backward) are cut into chunks: the same sums in the same
    order as the one-workgroup form -- checked EXACTLY against torch evaluating the same kw = 0 .. KS-1 sum in fp32"""
    k = K()
    N, H, Co, KS = 2, 3, 4, 9
    NP, half = KS * Co, KS // 2
    g = torch.Generator().manual_seed(41 + W)
    P = torch.randn(N, H, W, NP, generator=g)
    bias = torch.randn(Co, generator=g)
    s = bias.view(1, 1, 1, Co).expand(N, H, W, Co).clone()
    s0 = torch.zeros(N, H, W, Co)
    Pp = F.pad(P.view(N, H, W, KS, Co), (0, 0, 0, 0, half, half))            # zero columns either side of the row
    for kw in range(KS):                                                       # s[w] += P[w + kw - half][kw]
        s = s + Pp[:, :, kw:kw + W, kw, :]
        s0 = s0 + Pp[:, :, kw:kw + W, kw, :]
    Pd, bd = P.to(DEV), bias.to(DEV)
    out = torch.empty(N, Co, H, W, device=DEV)
    k.tail_shiftsum_tanh(Pd, bd, N, H, W, Co, KS, out)
    nh = torch.empty(N, H, W, Co, device=DEV)
    k.shiftsum_nhwc(Pd, N, H, W, Co, KS, nh)
    torch.cuda.synchronize()
    assert torch.equal(nh.cpu(), s0)                                           # same addends, same order: same bits
    assert relerr(out.cpu(), torch.tanh(s.double()).permute(0, 3, 1, 2)) < 2e-6
    # backward: dP[n][h][x][kw Co + co] = dpre[n][co][h][x - kw + half]; the bias-gradient partials add up to dpre's sum
    dout = torch.randn(N, Co, H, W, generator=g).to(DEV)
    nblk = k.tail_bwd_blocks(N, H, W, Co, KS)
    dP = torch.empty(N, H, W, NP, device=DEV); dbp = torch.empty(nblk, Co, device=DEV)
    k.tail_bwd(out, dout, N, H, W, Co, KS, dP, dbp, nblk)
    db = torch.zeros(Co, device=DEV)
    k.reduce_partials(dbp, nblk, Co, db, accumulate=False)
    torch.cuda.synchronize()
    dpre = (dout * (1.0 - out * out)).cpu()                                    # same fp32 expression as the kernel
    dpp = F.pad(dpre.permute(0, 2, 3, 1), (0, 0, half, half))                  # [N][H][W + 2 half][Co]
    want = torch.stack([dpp[:, :, 2 * half - kw:2 * half - kw + W, :] for kw in range(KS)], dim=3).reshape(N, H, W, NP)
    assert relerr(dP.cpu(), want) < 1e-6
    assert relerr(db.cpu(), dpre.double().sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("act", ["none", "mish", "relu"])
def test_bn_backward(act):
    k = K()
    M, C = 2 * 16 * 64, 64
    g = torch.Generator().manual_seed(31)
    y = torch.randn(M, C, generator=g) * 1.5 + 0.3
    gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g) * 0.2
    da = torch.randn(M, C, generator=g); da2 = torch.randn(M, C, generator=g)
    yr = y.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    z = F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)
    a = mish(z) if act == "mish" else (F.relu(z) if act == "relu" else z)
    a.backward((da + da2).double())
    mean = y.double().mean(0); var = y.double().var(0, unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    scale = (gamma.double() * rstd).float().to(DEV); shift = (beta.double() - mean * gamma.double() * rstd).float().to(DEV)
    sm, sr = mean.float().to(DEV), rstd.float().to(DEV)
    nblk = 37
    part = torch.empty(nblk, 2, C, device=DEV)
    yd, dad, da2d = y.to(DEV), da.to(DEV), da2.to(DEV)
    k.bn_bwd_reduce(dad, da2d, yd, M, C, scale, shift, sm, sr, act, part, nblk)
    dg = torch.zeros(C, device=DEV); dbt = torch.zeros(C, device=DEV); coef = torch.empty(3, C, device=DEV)
    gmd = gamma.to(DEV)
    k.bn_bwd_finalize(part, nblk, C, M, gmd, sm, sr, dg, dbt, coef, accumulate=False)
    dy = torch.empty(M, C, device=DEV)
    k.bn_bwd_apply(dad, da2d, yd, M, C, scale, shift, act, coef, dy)
    torch.cuda.synchronize()
    assert relerr(dg.cpu(), gr.grad) < 2e-5 and relerr(dbt.cpu(), br.grad) < 2e-5
    assert relerr(dy.cpu(), yr.grad) < 5e-5, relerr(dy.cpu(), yr.grad)


@pytest.mark.parametrize("M,C,nblk,two", [(1000, 256, 7, False), (130, 512, 1, True), (3 * 64 + 5, 64, 3, False), (4096, 32, 64, True)])
def test_bn_backward_reduce_shapes(M, C, nblk, two):
    """the statistics pass alone (rows per block not a multiple of the 4-row unroll, one / two gradient operands, 4..128 channel
    quads): per-channel sum dz and sum dz * xhat vs fp64"""
    k = K()
    g = torch.Generator().manual_seed(M + C)
    y = torch.randn(M, C, generator=g) * 1.5 + 0.3
    da = torch.randn(M, C, generator=g)
    da2 = torch.randn(M, C, generator=g) if two else None
    sc = torch.rand(C, generator=g) + 0.5; sh = torch.randn(C, generator=g) * 0.2
    mean = y.double().mean(0); rstd = 1 / torch.sqrt(y.double().var(0, unbiased=False) + 1e-5)
    z = y.double() * sc.double() + sh.double()
    sp = F.softplus(z); t = torch.tanh(sp)
    dmish = t + z * (1 - t * t) * torch.sigmoid(z)
    dz = (da.double() + (da2.double() if two else 0.0)) * dmish
    ref = torch.stack([dz.sum(0), (dz * (y.double() - mean) * rstd).sum(0)])
    part = torch.full((nblk, 2, C), float("nan"), device=DEV)
    k.bn_bwd_reduce(da.to(DEV), da2.to(DEV) if two else None, y.to(DEV), M, C, sc.to(DEV), sh.to(DEV), mean.float().to(DEV),
                    rstd.float().to(DEV), "mish", part, nblk)
    torch.cuda.synchronize()
    got = part.double().sum(0).cpu()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-5


@pytest.mark.parametrize("pool", [(2, 2), (1, 2), (1, 1)])
def test_affine_act_pool(pool):
    k = K()
    N, H, W, C = 3, 4, 10, 32
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, C, H, W, generator=g)
    sc = torch.randn(C, generator=g); sh = torch.randn(C, generator=g) * 0.2
    xr = x.double().requires_grad_(True)
    z = xr * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    zr = z.detach().requires_grad_(True)
    o = F.max_pool2d(F.relu(zr), pool, pool)
    do = torch.randn(o.shape, generator=g)
    o.backward(do.double())
    OH, OW = H // pool[0], W // pool[1]
    out = torch.empty(N * OH * OW, C, device=DEV)
    xd = to_nhwc(x)
    scd, shd, dod = sc.to(DEV), sh.to(DEV), to_nhwc(do)
    k.affine_act_pool(xd, N, H, W, C, scd, shd, "relu", pool[0], pool[1], out)
    dz = torch.full((N * H * W, C), float("nan"), device=DEV)
    k.affine_act_pool_bwd(xd, dod, N, H, W, C, scd, shd, "relu", pool[0], pool[1], dz)
    torch.cuda.synchronize()
    assert relerr(from_nhwc(out, N, OH, OW, C), o.detach()) < 1e-6
    assert relerr(from_nhwc(dz, N, H, W, C), zr.grad) < 1e-6


def test_prelu_add_transposes():
    k = K()
    g = torch.Generator().manual_seed(43)
    n = 4 * 1000
    x = torch.randn(n, generator=g); dy = torch.randn(n, generator=g); dy2 = torch.randn(n, generator=g)
    al = torch.tensor([0.23])
    xr = x.double().requires_grad_(True); ar = al.double().requires_grad_(True)
    y = F.prelu(xr, ar)
    y.backward((dy + dy2).double())
    yd = torch.empty(n, device=DEV)
    xd, ald, dyd, dy2d = x.to(DEV), al.to(DEV), dy.to(DEV), dy2.to(DEV)
    k.prelu_fwd(xd, ald, n, yd)
    dx = torch.empty(n, device=DEV); nblk = 16; dap = torch.empty(nblk, device=DEV); da = torch.zeros(1, device=DEV)
    k.prelu_bwd(xd, ald, dyd, dy2d, n, dx, dap, nblk)
    k.reduce_partials(dap, nblk, 1, da, accumulate=False)
    s = torch.empty(n, device=DEV)
    k.add(xd, dyd, n, s)
    t = torch.randn(2, 5, 3, 7, generator=g)
    a = torch.empty(2 * 3 * 7 * 5, device=DEV); bb = torch.empty(2 * 5 * 3 * 7, device=DEV)
    td = t.to(DEV)
    k.nchw_to_nhwc(td, 2, 5, 3, 7, a)
    k.nhwc_to_nchw(a, 2, 5, 3, 7, bb)
    torch.cuda.synchronize()
    assert relerr(yd.cpu(), y.detach()) < 1e-7 and relerr(dx.cpu(), xr.grad) < 1e-7 and relerr(da.cpu(), ar.grad) < 1e-5
    assert torch.equal(s.cpu(), x + dy)
    assert torch.equal(a.cpu().reshape(2, 3, 7, 5), t.permute(0, 2, 3, 1).contiguous()) and torch.equal(bb.cpu().reshape(t.shape), t)


@pytest.mark.parametrize("axis,N,H,W", [(0, 2, 5, 24), (1, 2, 8, 7), (0, 3, 16, 64), (1, 3, 16, 64)])
def test_bigru_fwd_bwd(axis, N, H, W):
    """Fused BiGRU vs the oracle's explicit gate equations + autograd (inputs: precomputed gi)."""
    k = K()
    g = torch.Generator().manual_seed(51 + axis)
    sd = O.recipe_state_dict(O._gru_spec("g", 64, 32), 77)
    w_ih = torch.cat([sd["g.weight_ih_l0"], sd["g.weight_ih_l0_reverse"]], 0)     # [192][64]
    b_ih = torch.cat([sd["g.bias_ih_l0"], sd["g.bias_ih_l0_reverse"]], 0)
    w_hh = torch.stack([sd["g.weight_hh_l0"], sd["g.weight_hh_l0_reverse"]], 0)   # [2][96][32]
    b_hh = torch.stack([sd["g.bias_hh_l0"], sd["g.bias_hh_l0_reverse"]], 0)
    u = torch.randn(N, H, W, 64, generator=g)                                      # NHWC
    ur = u.double().requires_grad_(True)
    p = {kk: v.double().requires_grad_(True) for kk, v in sd.items()}
    seq = ur if axis == 0 else ur.permute(0, 2, 1, 3)                              # (N, lines, T, C)
    B = seq.shape[0] * seq.shape[1]
    y = O.gru_bidir_explicit(seq.reshape(B, seq.shape[2], 64), p, "g").reshape(seq.shape[0], seq.shape[1], seq.shape[2], 64)
    y = y if axis == 0 else y.permute(0, 2, 1, 3)                                  # back to NHWC
    dh = torch.randn(N, H, W, 64, generator=g); dh2 = torch.randn(N, H, W, 64, generator=g)
    y.backward((dh + dh2).double())
    Pn = N * H * W
    gi = (u.reshape(Pn, 64) @ w_ih.t() + b_ih).to(DEV).contiguous()
    h_out = torch.full((Pn, 64), float("nan"), device=DEV)
    whd, bhd = w_hh.to(DEV).contiguous(), b_hh.to(DEV).contiguous()
    dhd, dh2d = dh.reshape(Pn, 64).to(DEV), dh2.reshape(Pn, 64).to(DEV)
    gates = torch.full((Pn, 256), float("nan"), device=DEV)
    h_inf = torch.empty(Pn, 64, device=DEV)
    k.bigru_fwd(gi, whd, bhd, N, H, W, axis, h_inf)                 # inference form: no gate store
    k.bigru_fwd(gi, whd, bhd, N, H, W, axis, h_out, gates)
    torch.cuda.synchronize()
    assert torch.equal(h_inf, h_out) and not torch.isnan(gates).any()
    assert (h_out.cpu().double() - y.detach().reshape(Pn, 64)).abs().max() < 2e-6
    dgi = torch.full((Pn, 192), float("nan"), device=DEV); dgh = torch.full((Pn, 192), float("nan"), device=DEV)
    k.bigru_bwd(gates, h_out, dhd, dh2d, whd, N, H, W, axis, dgi, dgh)
    torch.cuda.synchronize()
    dgi_c, dgh_c = dgi.cpu().double(), dgh.cpu().double()
    # input-side grads follow from dgi
    du = dgi_c @ w_ih.double()
    assert relerr(du, ur.grad.reshape(Pn, 64)) < 2e-5
    dwih = dgi_c.t() @ u.reshape(Pn, 64).double()
    ref_dwih = torch.cat([p["g.weight_ih_l0"].grad, p["g.weight_ih_l0_reverse"].grad], 0)
    assert relerr(dwih, ref_dwih) < 2e-5
    assert relerr(dgi_c.sum(0), torch.cat([p["g.bias_ih_l0"].grad, p["g.bias_ih_l0_reverse"].grad])) < 2e-5
    assert relerr(dgh_c.sum(0), torch.cat([p["g.bias_hh_l0"].grad, p["g.bias_hh_l0_reverse"].grad])) < 2e-5
    # hidden-side: dW_hh[d] = dgh[:, d]^T @ h_prev(d) via the wgrad GEMM with a one-step spatial shift
    for d, suf in ((0, ""), (1, "_reverse")):
        sgn = 1 if d == 0 else -1
        geom = k.ConvGeom(N, H, W, 32, 96, 1, 1, sgn if axis == 1 else 0, sgn if axis == 0 else 0, H, W)
        Z = k.wgrad_splits(geom.M, geom.K, 96)
        part = torch.empty(Z, 32, 96, device=DEV)
        ca = k.make_conv_args(geom, h_out, in_ld=64, in_coff=32 * d)
        k.conv_wgrad(k.make_wgrad_args(ca, dgh, part, None, dy_ld=192, dy_coff=96 * d))
        dw = torch.zeros(96, 32, device=DEV)
        k.wgrad_reduce(part, None, Z, geom, dw, None, accumulate=False)
        torch.cuda.synchronize()
        assert relerr(dw.cpu(), p["g.weight_hh_l0" + suf].grad) < 2e-5, (d, relerr(dw.cpu(), p["g.weight_hh_l0" + suf].grad))


def test_tps_and_grid_sample(golden_dir):
    import os
    k = K()
    gz = np.load(os.path.join(golden_dir, "op_tps.npz"))
    img, ctrl, gy = torch.tensor(gz["img"]), torch.tensor(gz["ctrl"]), torch.tensor(gz["gy"])
    N, C, H, W = img.shape
    tb = O.tps_buffers(H, W)
    inv, rep = tb["inverse_kernel"].to(DEV).contiguous(), tb["target_coordinate_repr"].to(DEV).contiguous()
    grid = torch.empty(N, H * W, 2, device=DEV); src = torch.empty(N, H * W, 2, device=DEV)
    ctrld = ctrl.to(DEV).contiguous()
    k.tps_grid_fwd(ctrld, inv, rep, N, H * W, 20, grid, src)
    out = torch.empty(N * H * W, C, device=DEV)
    xin = to_nhwc(img)
    k.grid_sample_fwd(xin, grid, N, H, W, C, H, W, False, out)
    torch.cuda.synchronize()
    assert (src.cpu() - torch.tensor(gz["src"])).abs().max() < 5e-5   # fp32 23-term dot products with O(10) kernel entries
    # the sampler itself is exact against ATen on the SAME grid; against the reference's own grid the 1e-5-level
    # fp32 difference of the TPS coordinates is amplified by W=64 pixels times the image gradient
    assert (from_nhwc(out, N, H, W, C) - F.grid_sample(img, grid.cpu().reshape(N, H, W, 2), align_corners=False)).abs().max() < 2e-5
    assert (from_nhwc(out, N, H, W, C) - torch.tensor(gz["y"])).abs().max() < 2e-3
    din = torch.empty(N * H * W, C, device=DEV); dgrid = torch.empty(N, H * W, 2, device=DEV); dctrl = torch.empty(N, 20, 2, device=DEV)
    gyd = to_nhwc(gy)
    k.grid_sample_bwd(xin, grid, gyd, N, H, W, C, H, W, False, din, dgrid)
    k.tps_grid_bwd(dgrid, src, inv, rep, N, H * W, 20, dctrl)
    torch.cuda.synchronize()
    assert relerr(from_nhwc(din, N, H, W, C), torch.tensor(gz["dimg"])) < 2e-3
    assert relerr(dctrl.cpu(), torch.tensor(gz["dctrl"])) < 5e-3, relerr(dctrl.cpu(), torch.tensor(gz["dctrl"]))
    # exact check of both backward kernels against ATen autograd on the same grid / src
    gr = grid.cpu().reshape(N, H, W, 2).clone().requires_grad_(True)
    ir = img.clone().requires_grad_(True)
    F.grid_sample(ir, gr, align_corners=False).backward(gy)
    assert relerr(from_nhwc(din, N, H, W, C), ir.grad) < 1e-5
    assert relerr(dgrid.cpu().reshape(N, H, W, 2), gr.grad) < 1e-5
    # the image gradient is a gather in a fixed order since round 5 (it was the library's last float atomicAdd): bitwise repeatable,
    # also next to a co-running kernel, and it overwrites whatever was in the buffer (no memset, no accumulation)
    first = din.clone()
    side = torch.cuda.Stream()
    junk = torch.randn(1 << 22, device=DEV)
    for it in range(50):
        din.fill_(float(it))
        with torch.cuda.stream(side):
            junk.mul_(1.0001)
        k.grid_sample_bwd(xin, grid, gyd, N, H, W, C, H, W, False, din, None)
        assert torch.equal(din, first), it
    torch.cuda.synchronize()
    # align_corners=True (the authors' torch 1.2 behaviour) against ATen
    ref = F.grid_sample(img, grid.cpu().reshape(N, H, W, 2), align_corners=True)
    k.grid_sample_fwd(xin, grid, N, H, W, C, H, W, True, out)
    torch.cuda.synchronize()
    assert (from_nhwc(out, N, H, W, C) - ref).abs().max() < 2e-5


def test_image_loss_kernels(golden_dir):
    import os
    k = K()
    gz = np.load(os.path.join(golden_dir, "losses.npz"))
    a, b = torch.tensor(gz["a"]).to(DEV), torch.tensor(gz["b"]).to(DEV)
    N, C, H, W = a.shape
    nblk = 64
    part = torch.empty(nblk, 2, device=DEV); loss = torch.empty(1, device=DEV)
    k.image_loss_fwd(a, b, N, C, H, W, True, part, nblk)
    k.image_loss_finalize(part, nblk, a.numel(), N * 3 * H * W, 1.0, 1e-4, loss)
    dl = torch.ones(1, device=DEV); da = torch.empty_like(a)
    k.image_loss_bwd(a, b, dl, N, C, H, W, True, 1.0, 1e-4, da)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(gz["image_loss"])) < 1e-6 * max(1, abs(float(gz["image_loss"])))
    assert relerr(da.cpu(), torch.tensor(gz["image_loss_grad"])) < 1e-5


def test_clip_and_adam():
    k = K()
    g = torch.Generator().manual_seed(61)
    n = 100003
    p0 = torch.randn(n, generator=g); grads = [torch.randn(n, generator=g) * 3 for _ in range(3)]
    pr = p0.clone(); opt = O.AdamState([pr])
    pd = p0.to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV); nblk = 128
    part = torch.empty(nblk, device=DEV); coef = torch.empty(1, device=DEV); nrm = torch.empty(1, device=DEV)
    for gr in grads:
        gc = gr.clone()
        tot = O.clip_grad_norm_([gc], 0.25)
        opt.step([gc])
        gd = gr.to(DEV)
        k.sumsq_partial(gd, n, part, nblk)
        k.clip_coef(part, nblk, 0.25, coef, nrm)
        k.step_inc(step)
        k.adam_step(pd, gd, m, v, n, coef, 1e-3, 0.5, 0.999, 1e-8, step)
        torch.cuda.synchronize()
        assert abs(nrm.item() - tot.item()) < 1e-4 * tot.item()
        assert (pd.cpu() - pr).abs().max() < 2e-6


@pytest.mark.parametrize("N,H,dgrad,bias,bn,concat", [(2, 5, False, True, True, False), (1, 16, True, False, False, False),
                                                       (3, 7, False, True, True, True), (2, 3, True, True, False, True)])
def test_conv3x3_wstat_kernel_direct(N, H, dgrad, bias, bn, concat, monkeypatch):
    """The weights-stationary kernel of the 64-channel 3x3 trunk convs in isolation: N*H not divisible by
    3 (ragged last workgroup), concat-style in_ld / in_coff and out_ld / out_coff, with and without bias and BN partials,
    forward and data-gradient geometry -- against the fp64 restatement AND against the tile-loop kernel (TPGSR_CONV_WSTAT
    is read once per process, so the comparison kernel is reached through a shape the fast path rejects: a residual add of zeros)."""
    from tpgsr_amd import kernels as K
    W, C = 64, 64
    g = torch.Generator().manual_seed(N * 100 + H)
    in_ld, in_coff = (96, 32) if concat else (64, 0)
    out_ld, out_coff = (128, 64) if concat else (64, 0)
    xfull = torch.randn(N * H * W, in_ld, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    b = torch.randn(C, generator=g) if bias else None
    x = xfull[:, in_coff:in_coff + C].reshape(N, H, W, C).permute(0, 3, 1, 2).double()
    if not dgrad:
        ref = F.conv2d(x, w.double(), None, padding=1)
        wt = w.permute(2, 3, 1, 0).reshape(9 * C, C).contiguous()                 # [(kh*3+kw)*Cin+ci][co]
    else:   # data gradient of a conv with weight w: conv of dy with the flipped, transposed taps
        ref = F.conv_transpose2d(x, w.double(), None, padding=1)
        wt = w.flip(2, 3).permute(2, 3, 0, 1).reshape(9 * C, C).contiguous()      # [((2-kh)*3+(2-kw))*Cout+co][ci]
    ref_nb = ref.permute(0, 2, 3, 1).reshape(-1, C)
    xd, wd = xfull.to(DEV), wt.to(DEV)
    bd = b.to(DEV) if bias else None
    geom = K.ConvGeom(N, H, W, C, C, 3, 3, 1, 1)
    outs = []
    for force_tile_loop in (False, True):
        out = torch.full((N * H * W, out_ld), 9.0, device=DEV)
        nblk = (N * H * W + 63) // 64
        part = torch.zeros(nblk, 2, C, device=DEV) if bn else None
        kw = dict(bias=bd, in_ld=in_ld, in_coff=in_coff, out_ld=out_ld, out_coff=out_coff, bn_partial=part)
        if force_tile_loop:
            kw["in2"] = torch.zeros(N * H * W, C, device=DEV)
        K.conv_fwd(K.make_conv_args(geom, xd, wd, out, **kw))
        torch.cuda.synchronize()
        got = out[:, out_coff:out_coff + C].cpu().double()
        want = ref_nb + (b.double() if bias else 0.0)
        err_rows = (got - want).abs().reshape(N * H, W * C).max(1).values
        print(f"wstat test N={N} H={H} dgrad={dgrad} tile_loop={force_tile_loop}: per-image-row max err", [f"{v:.1e}" for v in err_rows.tolist()])
        assert (got - want).abs().max() < 2e-4 * max(1.0, want.abs().max().item()), ("tile loop" if force_tile_loop else "wstat kernel")
        if concat:
            assert (out[:, :out_coff] == 9.0).all()            # neighbouring channels untouched
        if bn:
            s = part.sum(0).cpu().double()
            assert (s[0] - ref_nb.sum(0)).abs().max() < 1e-3 * ref_nb.abs().sum(0).max()
            assert (s[1] - (ref_nb ** 2).sum(0)).abs().max() < 1e-3 * (ref_nb ** 2).sum(0).max()
        outs.append((got, part.clone() if bn else None))
    assert (outs[0][0] - outs[1][0]).abs().max() < 2e-5 * max(1.0, outs[1][0].abs().max().item())
    if bn:   # same [row block][2][C] layout from both kernels
        assert (outs[0][1] - outs[1][1]).abs().max() < 1e-3 * outs[1][1].abs().max()
