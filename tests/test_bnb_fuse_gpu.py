"""GPU: BatchNorm-backward statistics in the epilogue of the convolution that produces the incoming gradient (tpgsr_conv_args.bnb_y,
csrc/conv_xbf_common.h) against an fp64 restatement of autograd's reduction (model/tsrn.py:376,380: BatchNorm2d [+ mish] in training
mode; sum dz and sum dz * xhat per channel, dz = da * act'(scale * y + shift)), per 64-pixel row block, on every kernel family that
carries the epilogue (halo, row-panel, tile loop), in the three split arithmetics; then the whole train step with the fusion on / off."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _act_grad64(z, act):
    if act == "none":
        return torch.ones_like(z)
    if act == "relu":
        return (z > 0).double()
    sp = F.softplus(z)
    t = torch.tanh(sp)
    return t + z * (1 - t * t) * torch.sigmoid(z)      # d/dz z tanh(softplus z)


def _case(N, H, W, Ci, Co, KH, KW, ph, pw, act, terms, seed, in_ps=False):
    """data-gradient-like launch dy [M][Ci] -> da [M][Co] with the BatchNorm input y [M][Co]; returns the max relative errors of
    (da, per-block sum dz, per-block sum dz xhat) and the kernel's partial tensor"""
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, KH, KW, generator=g) / math.sqrt(Ci * KH * KW)
    geom = K.ConvGeom(N, H, W, Ci, Co, KH, KW, ph, pw)
    M = geom.M
    y = torch.randn(M, Co, generator=g) * 1.5 + 0.3
    mean, rstd = torch.randn(Co, generator=g) * 0.2 + 0.3, torch.rand(Co, generator=g) + 0.4
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.3
    da64 = F.conv2d(x.double(), w.double(), None, padding=(ph, pw)).permute(0, 2, 3, 1).reshape(M, Co)
    dz = da64 * _act_grad64(y.double() * sc.double() + sh.double(), act)
    xhat = (y.double() - mean.double()) * rstd.double()
    pad = (-M) % 64
    blk = lambda t: torch.cat([t, t.new_zeros(pad, Co)]).reshape(-1, 64, Co).sum(1)
    ref_s, ref_sx = blk(dz), blk(dz * xhat)
    with K.conv_terms(terms):
        wf = w.permute(2, 3, 1, 0).reshape(KH * KW * Ci, Co).contiguous().to(DEV)
        K.make_bf_twin(wf, Ci)
        if in_ps:     # the operand stored pixel-shuffled, [N][2H][2W][Ci/4] (the upsample block's data gradient)
            xin = F.pixel_shuffle(x, 2).permute(0, 2, 3, 1).reshape(-1, Ci // 4).contiguous().to(DEV)
        else:
            xin = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous().to(DEV)
        out = torch.full((M, Co), float("nan"), device=DEV)
        part = torch.full(((M + 63) // 64, 2, Co), float("nan"), device=DEV)
        keep = [y.to(DEV), mean.to(DEV), rstd.to(DEV), sc.to(DEV), sh.to(DEV)]
        bnb = dict(y=keep[0], mean=keep[1], rstd=keep[2], scale=keep[3], shift=keep[4], act=act, partial=part)
        K.conv_fwd(K.make_conv_args(geom, xin, wf, out, in_ps=in_ps, bnb=bnb))
        torch.cuda.synchronize()
    e_out = ((out.cpu().double() - da64).abs().max() / da64.abs().max()).item()
    p = part.cpu().double()
    e_s = ((p[:, 0] - ref_s).abs().max() / ref_s.abs().max()).item()
    e_sx = ((p[:, 1] - ref_sx).abs().max() / ref_sx.abs().max()).item()
    return e_out, e_s, e_sx, (out, part, keep)


TOL = {3: 3e-6, 2: 4e-5, 1: 2e-2}

SHAPES = [
    # N, H, W, Ci, Co, KH, KW, ph, pw, act            kernel family under the default routing
    (48, 16, 64, 64, 64, 3, 3, 1, 1, "mish"),       # halo (trunk conv2 data gradient -> bn1 + mish): 768 tiles on a persistent grid
    (48, 16, 64, 192, 64, 1, 1, 0, 0, "none"),      # row panel (GruBlock projection data gradient -> bn2)
    (5, 8, 25, 128, 96, 3, 3, 1, 1, "relu"),        # halo, tiles spanning rows and images, Cout not a multiple of 64 (recognizer)
    (3, 5, 13, 192, 64, 1, 1, 0, 0, "none"),        # tile loop (195 pixels: below the panel threshold), ragged last row block
    (3, 5, 13, 64, 40, 3, 3, 1, 1, "mish"),         # halo, ragged M and ragged Cout
    (2, 16, 50, 128, 64, 3, 3, 1, 1, "relu"),       # tile loop (halo of 278 entries: the halo kernel declines), recognizer conv1 gradient
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("terms", [3, 2, 1])
def test_bnb_epilogue_vs_fp64(shape, terms):
    e_out, e_s, e_sx, _ = _case(*shape, terms=terms, seed=13)
    print(f"bnb {shape} T={terms}: da {e_out:.2e}  sum dz {e_s:.2e}  sum dz xhat {e_sx:.2e}")
    assert e_out < TOL[terms] and e_s < 8 * TOL[terms] and e_sx < 8 * TOL[terms]


def test_bnb_epilogue_unshuffle_gather():
    """data gradient of the upsample convolution (256 -> 64 through the un-PixelShuffle gather, LD 8) -> bn7, no activation"""
    e_out, e_s, e_sx, _ = _case(6, 16, 64, 256, 64, 3, 3, 1, 1, "none", terms=2, seed=3, in_ps=True)
    print(f"bnb un-PixelShuffle: da {e_out:.2e}  sum dz {e_s:.2e}  sum dz xhat {e_sx:.2e}")
    assert e_out < TOL[2] and e_s < 8 * TOL[2] and e_sx < 8 * TOL[2]


@pytest.mark.parametrize("with_sums", [False, True])
def test_epilogue_stores_activation_backward(with_sums):
    """bnb_store_dz: `out` receives da * act'(y [* scale + shift]) -- without bn_partial a plain activation backward on the way out
    (the mish in front of the tail convolution: model/tsrn.py:39,159), with it the sums as well"""
    from tpgsr_amd import kernels as K
    g = torch.Generator().manual_seed(9)
    N, H, W, Ci, Co, KH, KW = 3, 32, 128, 36, 64, 9, 1          # the tail's data gradient: 9x1 over the folded 36 columns
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, KH, KW, generator=g) / math.sqrt(Ci * KH * KW)
    geom = K.ConvGeom(N, H, W, Ci, Co, KH, KW, 4, 0)
    M = geom.M
    y = torch.randn(M, Co, generator=g) * 1.5
    mean, rstd = torch.randn(Co, generator=g) * 0.2, torch.rand(Co, generator=g) + 0.4
    da64 = F.conv2d(x.double(), w.double(), None, padding=(4, 0)).permute(0, 2, 3, 1).reshape(M, Co)
    dz64 = da64 * _act_grad64(y.double(), "mish")
    with K.conv_terms(2):
        wf = w.permute(2, 3, 1, 0).reshape(KH * KW * Ci, Co).contiguous().to(DEV)
        K.make_bf_twin(wf, Ci)
        xin = x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous().to(DEV)
        out = torch.full((M, Co), float("nan"), device=DEV)
        keep = [y.to(DEV), mean.to(DEV), rstd.to(DEV)]
        bnb = dict(y=keep[0], act="mish", store_dz=True)
        part = None
        if with_sums:
            part = torch.full(((M + 63) // 64, 2, Co), float("nan"), device=DEV)
            one, zero = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
            bnb.update(partial=part, mean=keep[1], rstd=keep[2], scale=one, shift=zero)
        K.conv_fwd(K.make_conv_args(geom, xin, wf, out, bnb=bnb))
        torch.cuda.synchronize()
    assert ((out.cpu().double() - dz64).abs().max() / dz64.abs().max()).item() < TOL[2]
    if with_sums:
        ref = (dz64 * (y.double() - mean.double()) * rstd.double()).reshape(-1, 64, Co).sum(1)
        assert ((part[:, 1].cpu().double() - ref).abs().max() / ref.abs().max()).item() < 8 * TOL[2]


def test_bnb_epilogue_equals_reduce_kernel_totals():
    """the fused sums against tpgsr_bn_bwd_reduce on the SAME stored gradient: what tpgsr_bn_bwd_finalize sees is the same to fp32
    rounding of a different summation order"""
    from tpgsr_amd import kernels as K
    N, H, W, C = 8, 16, 64, 64
    _, _, _, (da, part, keep) = _case(N, H, W, 64, C, 3, 3, 1, 1, "mish", terms=2, seed=21)
    M = N * H * W
    nblk = min(1024, M // 64)
    p2 = torch.empty(nblk, 2, C, device=DEV)
    K.bn_bwd_reduce(da, None, keep[0], M, C, keep[3], keep[4], keep[1], keep[2], "mish", p2, nblk)
    torch.cuda.synchronize()
    a, b = part.double().sum(0), p2.double().sum(0)
    assert ((a - b).abs().max() / b.abs().max()).item() < 2e-6


def test_bnb_rejected_on_fp32_kernel():
    from tpgsr_amd import kernels as K
    with K.conv_terms(0):
        x = torch.randn(128, 64, device=DEV)
        w = torch.randn(64, 64, device=DEV)
        with pytest.raises(RuntimeError, match="split-bf16"):
            K.make_conv_args(K.ConvGeom(2, 1, 64, 64, 64), x, w, torch.empty(128, 64, device=DEV),
                             bnb=dict(y=x, mean=w[0], rstd=w[1], scale=w[2], shift=w[3], act="none", partial=torch.empty(2, 2, 64, device=DEV)))
        # and the C entry point refuses a hand-built argument block that would land on the fp32 kernel
        a = K.make_conv_args(K.ConvGeom(2, 1, 64, 64, 64), x, w, torch.empty(128, 64, device=DEV))
        a.bnb_y, a.bnb_mean, a.bnb_rstd, a.bn_partial = x.data_ptr(), w.data_ptr(), w.data_ptr(), torch.empty(2, 2, 64, device=DEV).data_ptr()
        with pytest.raises(RuntimeError, match="split-bf16"):
            K.conv_fwd(a)


def _c3_grads(fuse: bool):
    """forward + backward of one TPGSR step (TSRN_TL + STN, teacher, one student; batch 4) -> the flat gradient arena"""
    from tpgsr_amd import kernels as K
    from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
    import test_crnn_gpu as TC
    old, K.BNB_FUSE = K.BNB_FUSE, fuse
    try:
        srs, stus, teacher, *_ = TC._c3_models()
        lr, hr = TC.O.synthetic_batch(4, 77)
        ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=1)
        ts.pool.bind(torch.device(DEV, 0))
        teacher._engine().bind(torch.device(DEV, 0))
        loss = ts._phase_a(lr.to(DEV), hr.to(DEV))
        torch.cuda.synchronize()
        return loss.item(), ts.pool.grad.clone(), ts.pool
    finally:
        K.BNB_FUSE = old


def test_train_step_gradients_fused_vs_unfused():
    """every gradient of the SR network and the student with the statistics riding on the producing convolutions (17 launches fewer
    at C3, plus the tail's activation backward) against the same step with tpgsr_bn_bwd_reduce as its own launch: equal up to the summation order of two sums per
    BatchNorm (fp32)"""
    la, ga, pool = _c3_grads(True)
    lb, gb, _ = _c3_grads(False)
    assert la == lb                                     # the forward pass does not change
    worst = 0.0
    for mod, (lo, hi) in pool.ranges.items():
        a, b = ga[lo:hi].double(), gb[lo:hi].double()
        rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        worst = max(worst, rel)
    print("fused vs unfused BatchNorm-backward sums: worst relative gradient difference per module", worst)
    assert worst < 2e-5
