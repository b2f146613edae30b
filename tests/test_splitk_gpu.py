"""GPU: split-K of the tile loop (tpgsr_conv_args.sk_splits, csrc/conv_xbf.hip, round 6) -- S workgroups per output tile over 1 / S of the
K chunks each + a reduce launch that runs the ordinary epilogue -- against the unsplit launch (same products, partial sums associated
differently: last-bit differences) and against fp64, on the shapes it is for (fewer 64 x 64 tiles than CUs, >= 24 K chunks: the BiLSTM
projections' data gradients, conv6, InfoGen's zero-dilated transposed convolutions, the STN head's convolutions on 96-pixel maps), with the
fused prologues, bias, BatchNorm statistics and the BatchNorm-backward epilogue; and the planner's refusals."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

SHAPES = [
    # N, H, W, Ci, Co, KH, KW, ph, pw, dil_w, what
    (48, 1, 26, 2048, 512, 1, 1, 0, 0, 1, "BiLSTM projection data gradient"),
    (48, 1, 26, 2048, 256, 1, 1, 0, 0, 1, "BiLSTM projection data gradient (256)"),
    (48, 2, 27, 512, 512, 2, 2, 0, 0, 1, "conv6"),
    (48, 1, 101, 512, 128, 1, 3, 0, 1, 2, "InfoGen transposed convolution over a zero-dilated strip"),
    (48, 1, 2, 256, 256, 3, 3, 1, 1, 1, "STN head convolution on 96 pixels"),
    (5, 3, 7, 96, 40, 3, 3, 1, 1, 1, "ragged: 105 pixels, 40 columns, 27 chunks"),
]


def _run(shape, terms, split, *, affine=False, act=False, bn=True, bias=True, bnb=False, seed=0):
    from tpgsr_amd import _lib, kernels as K
    N, H, W, Ci, Co, KH, KW, ph, pw, dil, _ = shape
    g = torch.Generator().manual_seed(seed)
    Wr = (W - 1) // dil + 1                       # real width of a zero-dilated input
    x = torch.randn(N * H * Wr, Ci, generator=g)
    sc = (torch.rand(Ci, generator=g) + 0.5) if affine else None
    sh = (torch.randn(Ci, generator=g) * 0.3) if affine else None
    w = torch.randn(KH * KW * Ci, Co, generator=g) / math.sqrt(Ci * KH * KW)
    b = torch.randn(Co, generator=g) if (bias and not bnb) else None
    geom = K.ConvGeom(N, H, W, Ci, Co, KH, KW, ph, pw)
    dev = lambda t: None if t is None else t.to(DEV).contiguous()
    keep = [dev(x), dev(sc), dev(sh), dev(b)]
    wf = w.to(DEV)
    lib = _lib.load()
    lib.tpgsr_splitk_set_enabled(1)
    S = 0
    try:
        with K.conv_terms(terms):
            K.make_bf_twin(wf, 0)
            out = torch.full((geom.M, Co), float("nan"), device=DEV)
            part = torch.full(((geom.M + 63) // 64, 2, Co), float("nan"), device=DEV) if (bn or bnb) else None
            if bnb:
                y = torch.randn(geom.M, Co, generator=g).to(DEV)
                mean, rstd = torch.randn(Co, generator=g).to(DEV) * 0.1, (torch.rand(Co, generator=g) + 0.5).to(DEV)
                bsc, bsh = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) * 0.2).to(DEV)
                keep += [y, mean, rstd, bsc, bsh]
                kw = dict(bnb=dict(y=y, mean=mean, rstd=rstd, scale=bsc, shift=bsh, act="mish", partial=part))
            else:
                kw = dict(bn_partial=part)
            a = K.make_conv_args(geom, keep[0], wf, out, bias=keep[3], in_scale=keep[1], in_shift=keep[2], in_act="relu" if act else None,
                                 in_dil_w=dil, **kw)
            if not split:
                a.sk_splits = 1                    # (0 would be offered to the planner by K.conv_fwd; 1 = explicitly unsplit)
            if split:
                nb = C.c_longlong(0)
                S = lib.tpgsr_conv_splitk_plan(C.byref(a), C.byref(nb))
                assert S > 1 and nb.value == S * ((geom.M + 63) // 64) * ((Co + 63) // 64) * 256 * 16 * 4, (S, nb.value)
                buf = torch.full((nb.value // 4,), float("nan"), device=DEV)
                keep.append(buf)
                a.sk_part, a.sk_splits = buf.data_ptr(), S
            K.conv_fwd(a)
        torch.cuda.synchronize()
    finally:
        lib.tpgsr_splitk_set_enabled(1)
    ref = None
    if not bnb:
        xa = x.double()
        if affine:
            xa = xa * sc.double() + sh.double()
        if act:
            xa = torch.relu(xa)
        x4 = xa.view(N, H, Wr, Ci).permute(0, 3, 1, 2)
        if dil > 1:
            z = torch.zeros(N, Ci, H, W, dtype=torch.float64)
            z[:, :, :, ::dil] = x4
            x4 = z
        w4 = w.double().view(KH, KW, Ci, Co).permute(3, 2, 0, 1)
        ref = F.conv2d(x4, w4, b.double() if b is not None else None, padding=(ph, pw)).permute(0, 2, 3, 1).reshape(-1, Co)
    return out, part, ref, S


@pytest.mark.parametrize("terms", [2, 3])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[-1] for s in SHAPES])
def test_splitk_equals_the_unsplit_launch_and_fp64(shape, terms):
    base, bpart, ref, _ = _run(shape, terms, False, affine=True, act=True)
    out, part, _, S = _run(shape, terms, True, affine=True, act=True)
    scale = ref.abs().max().item()
    d_split = (out - base).abs().max().item() / scale
    e_base = (base.cpu().double() - ref).abs().max().item() / scale
    e_split = (out.cpu().double() - ref).abs().max().item() / scale
    ps = bpart.abs().max().item()
    d_part = (part - bpart).abs().max().item() / ps
    print(f"{shape[-1]} x{terms}: S = {S}; split vs unsplit {d_split:.2e} (statistics {d_part:.2e}); vs fp64: unsplit {e_base:.2e}, split {e_split:.2e}")
    assert not torch.isnan(out).any() and not torch.isnan(part).any()
    assert d_split < 2e-6 and d_part < 2e-6
    assert e_split < max(2.0 * e_base, 1e-6)


@pytest.mark.parametrize("terms", [2, 3])
def test_splitk_with_the_batchnorm_backward_epilogue(terms):
    """the data gradient that enters a BatchNorm's backward pass (tpgsr_conv_args.bnb_*): dz and the two reduction sums come out of the
    reduce launch's epilogue"""
    shape = SHAPES[0]
    base, bpart, _, _ = _run(shape, terms, False, bnb=True)
    out, part, _, S = _run(shape, terms, True, bnb=True)
    d = (out - base).abs().max().item() / base.abs().max().item()
    dp = (part - bpart).abs().max().item() / bpart.abs().max().item()
    print(f"bnb epilogue x{terms}: S = {S}; dz {d:.2e}, sums {dp:.2e}")
    assert d < 2e-6 and dp < 5e-6


def test_splitk_is_repeatable_and_the_planner_declines_what_it_should():
    from tpgsr_amd import _lib, kernels as K
    a1, _, _, _ = _run(SHAPES[3], 2, True, affine=True, act=True)
    a2, _, _, _ = _run(SHAPES[3], 2, True, affine=True, act=True)
    assert torch.equal(a1, a2)
    lib = _lib.load()
    nb = C.c_longlong(0)
    x = torch.zeros(8, device=DEV)

    def plan(N, H, W, Ci, Co, KH=1, KW=1, p=0, terms=2):
        with K.conv_terms(terms):
            wf = torch.zeros(KH * KW * Ci, Co, device=DEV)
            K.make_bf_twin(wf, 0)
            a = K.make_conv_args(K.ConvGeom(N, H, W, Ci, Co, KH, KW, p, p), x, wf, x)
            return lib.tpgsr_conv_splitk_plan(C.byref(a), C.byref(nb))
    lib.tpgsr_splitk_set_enabled(0)
    assert plan(48, 1, 26, 2048, 512) == 0                         # switched off
    lib.tpgsr_splitk_set_enabled(1)
    try:
        assert plan(48, 1, 26, 2048, 512) == 4
        assert plan(48, 1, 26, 512, 2048) == 0                     # 640 tiles: enough workgroups
        assert plan(48, 1, 26, 256, 512) == 0                      # 8 chunks: too short
        assert plan(48, 16, 64, 64, 64, 3, 3, 1) == 0              # the trunk: the halo kernels' launch
        assert plan(48, 1, 26, 2048, 512, terms=0) == 0            # fp32 kernel
    finally:
        lib.tpgsr_splitk_set_enabled(1)
