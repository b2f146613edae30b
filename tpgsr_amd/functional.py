"""Operator-level API: every building block of the reference's networks as a differentiable function on the HIP kernels.

The fused engines (engine.py / engine_crnn.py) run whole networks as recorded plans.  This module is the same kernel
library exposed one operator at a time -- a `torch.autograd.Function` per op whose forward AND backward are C-ABI launches --
so that (i) the reference's block classes work standalone (`STNHead(x) -> (feat, ctrl)`, `TPSSpatialTransformer`,
`GruBlock`, `RecurrentResidualBlock(TL)`, `InfoGen`, `UpsampleBLock`, `mish`: SURVEY.md section 8b), and (ii) the other
`--arch` / `--tpg` choices of the reference (the _TL baseline backbones, the OPT text-prior generator) are written the way
the reference writes them, layer by layer, without a single ATen compute kernel.

Conventions: activations are fp32 CUDA tensors in NHWC order (N, H, W, C), contiguous; `to_nhwc` / `to_nchw` convert at the
module boundary.  Parameters keep the PyTorch layouts (state_dict interchange); weights are packed per call (these paths are
not the tuned hot path).  Nothing here falls back to a stock PyTorch kernel: non-CUDA inputs raise."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import kernels as K
from .kernels import ConvGeom

F32 = torch.float32


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda or K.DRYRUN):
            raise RuntimeError("tpgsr_amd.functional runs on the GPU only (no CPU / stock-PyTorch fallback)")
        if t.dtype != F32:
            raise TypeError(f"tpgsr_amd.functional expects fp32 tensors, got {t.dtype}")


def _new(like, *shape):
    return torch.empty(*shape, dtype=F32, device=like.device)


# Parameter-gradient sinks (engine_functional.py, recorded mode): data_ptr of a parameter -> the fp32 buffer its gradient is ACCUMULATED
# into by the operator's own backward kernels (the slab reduce / the BatchNorm-backward finalize with accumulate = 1).  The backward then
# hands autograd `None` for that parameter: no AccumulateGrad node runs, i.e. no ATen `add_` that a recorded plan could not replay.
GRAD_SINK: dict = {}


def _sink(p):
    return GRAD_SINK.get(p.data_ptr()) if (GRAD_SINK and p is not None) else None


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- layout ---------------------------------------------------------------------------------------------------------
class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        N, C, H, W = x.shape
        out = _new(x, N, H, W, C)
        K.nchw_to_nhwc(_c(x), N, C, H, W, out)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = g.shape
        out = _new(g, N, C, H, W)
        K.nhwc_to_nchw(_c(g), N, C, H, W, out)
        return out


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        N, H, W, C = x.shape
        out = _new(x, N, C, H, W)
        K.nhwc_to_nchw(_c(x), N, C, H, W, out)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = g.shape
        out = _new(g, N, H, W, C)
        K.nchw_to_nhwc(_c(g), N, C, H, W, out)
        return out


def to_nhwc(x):
    return _ToNHWC.apply(x)


def to_nchw(x):
    return _ToNCHW.apply(x)


# ---- convolution / linear ------------------------------------------------------------------------------------------
def _pack(w, transposed, wscale):
    """PyTorch conv weight -> (wt_f [K][Cout_p], wt_d [KH*KW*Cout][Cin_p]); channel counts that are not multiples of 4 are
    zero-padded in the operand's row length so the 16-byte weight loads apply"""
    if transposed:
        Cin, Cout, KH, KW = w.shape
    else:
        Cout, Cin, KH, KW = w.shape
    wt_f = torch.empty(KH * KW * Cin, Cout, dtype=F32, device=w.device)
    wt_d = torch.empty(KH * KW * Cout, Cin, dtype=F32, device=w.device)
    K.pack_conv_weight(_c(w), Cout, Cin, KH, KW, wt_f, wt_d, transposed=transposed, wscale=wscale)
    if K.CONV_TERMS:
        K.make_bf_twin(wt_f, Cin)
        K.make_bf_twin(wt_d, Cout)
    return wt_f, wt_d, Cout, Cin, KH, KW


class _Conv2d(torch.autograd.Function):
    """stride-1 conv (nn.Conv2d / nn.Linear as 1x1); `transposed`: the weight is a ConvTranspose2d weight [Cin][Cout][KH][KW]
    and the op is its equivalent stride-1 conv (flipped taps) over an already zero-dilated input."""

    @staticmethod
    def forward(ctx, x, w, b, pad_h, pad_w, out_ps, transposed, wscale):
        _chk(x, w, b)
        x = _c(x)
        N, H, W, Cx = x.shape
        wt_f, wt_d, Cout, Cin, KH, KW = _pack(w, transposed, wscale)
        if Cx != Cin:
            raise ValueError(f"conv2d: input has {Cx} channels, weight expects {Cin}")
        g = ConvGeom(N, H, W, Cin, Cout, KH, KW, pad_h, pad_w)
        if g.OH <= 0 or g.OW <= 0:
            raise ValueError("conv2d: empty output")
        out = _new(x, N, 2 * g.OH, 2 * g.OW, Cout // 4) if out_ps else _new(x, N, g.OH, g.OW, Cout)
        K.conv_fwd(K.make_conv_args(g, x, wt_f, out, bias=b, out_ps=out_ps))
        ctx.save_for_backward(x, w, wt_d)
        ctx.cfg = (g, out_ps, transposed, wscale, b is not None)
        ctx.bias_ptr = b.data_ptr() if b is not None else None     # (the sinks are looked up when the backward runs)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, wt_d = ctx.saved_tensors
        g, out_ps, transposed, wscale, has_b = ctx.cfg
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _new(x, g.N, g.H, g.W, g.Cin)
            K.conv_fwd(K.make_conv_args(g.dgrad(), dy, wt_d, dx, in_ps=out_ps))
        if ctx.needs_input_grad[1] or (has_b and ctx.needs_input_grad[2]):
            Z = K.wgrad_splits(g.M, g.K, g.Cout, geom=g)
            part = _new(x, Z, g.K, g.Cout)
            dbp = _new(x, Z, g.Cout) if has_b else None
            sw = _sink(w)
            with K.side():      # a leaf of the graph: on the weight-gradient stream when a recorded plan overlaps them (no-op eagerly)
                K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(g, x), dy, part, dbp, dy_ps=out_ps, zsplits=Z))
                if sw is not None:
                    sb = GRAD_SINK.get(ctx.bias_ptr) if has_b else None
                    assert not has_b or sb is not None, "a sunk weight gradient needs its bias gradient sunk too"
                    K.wgrad_reduce(part, dbp, Z, g, sw, sb, layout=1 if transposed else 0, accumulate=True, gscale=wscale)
                else:
                    dw = torch.empty_like(w)
                    db = _new(x, g.Cout) if has_b else None
                    K.wgrad_reduce(part, dbp, Z, g, dw, db, layout=1 if transposed else 0, accumulate=False, gscale=wscale)
        return dx, dw, db, None, None, None, None, None


def conv2d(x, w, b=None, padding=0, out_ps=False, wscale=1.0):
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    return _Conv2d.apply(x, w, b, ph, pw, bool(out_ps), False, float(wscale))


def linear(x, w, b=None, wscale=1.0):
    """x (..., Cin) -> (..., Cout) with an nn.Linear weight [Cout][Cin]"""
    lead = x.shape[:-1]
    y = conv2d(x.reshape(1, 1, -1, x.shape[-1]), w.reshape(w.shape[0], w.shape[1], 1, 1), b, 0, wscale=wscale)
    return y.reshape(*lead, w.shape[0])


class _Dilate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sh, sw):
        _chk(x)
        x = _c(x)
        N, H, W, C = x.shape
        out = _new(x, N, (H - 1) * sh + 1, (W - 1) * sw + 1, C)
        K.dilate2d(x, N, H, W, C, sh, sw, out)
        ctx.cfg = (sh, sw)
        return out

    @staticmethod
    def backward(ctx, g):
        sh, sw = ctx.cfg
        g = _c(g)
        N, H, W, C = g.shape
        out = _new(g, N, (H + sh - 1) // sh, (W + sw - 1) // sw, C)
        K.subsample2d(g, N, H, W, C, sh, sw, out)
        return out, None, None


class _Subsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sh, sw):
        _chk(x)
        x = _c(x)
        N, H, W, C = x.shape
        out = _new(x, N, (H + sh - 1) // sh, (W + sw - 1) // sw, C)
        K.subsample2d(x, N, H, W, C, sh, sw, out)
        ctx.cfg = (sh, sw, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        sh, sw, H, W = ctx.cfg
        g = _c(g)
        N, OH, OW, C = g.shape
        d = _new(g, N, (OH - 1) * sh + 1, (OW - 1) * sw + 1, C)
        K.dilate2d(g, N, OH, OW, C, sh, sw, d)
        if d.shape[1] == H and d.shape[2] == W:
            return d, None, None
        full = torch.zeros(N, H, W, C, dtype=F32, device=g.device)      # rows / columns past the last sample get no gradient
        _rows_copy(d, full)
        return full, None, None


def _rows_copy(d, full):
    N, dH, dW, C = d.shape
    _, H, W, _ = full.shape
    for n in range(N):
        K.copy_strided(d[n], dW * C, 0, full[n], W * C, 0, dH, dW * C)


def subsample(x, sh, sw):
    return x if (sh == 1 and sw == 1) else _Subsample.apply(x, sh, sw)


def conv_transpose2d(x, w, stride=1, padding=0):
    """nn.ConvTranspose2d(bias=False, output_padding=0): zero-dilate the input, then the equivalent stride-1 conv"""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    KH, KW = w.shape[2], w.shape[3]
    xd = x if (sh == 1 and sw == 1) else _Dilate.apply(x, sh, sw)
    return _Conv2d.apply(xd, w, None, KH - 1 - ph, KW - 1 - pw, False, True, 1.0)


def conv2d_strided(x, w, b=None, stride=(1, 1), padding=0):
    """strided conv = stride-1 conv + sub-sampling (only the small conv4_1 of the OPT feature extractor uses it)"""
    return subsample(conv2d(x, w, b, padding), stride[0], stride[1])


# ---- BatchNorm (+ fused activation) ----------------------------------------------------------------------------------
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, training, momentum, eps, act):
        _chk(x, gamma, beta)
        x = _c(x)
        C = x.shape[-1]
        M = x.numel() // C
        dev = x.device
        scale, shift = torch.empty(C, device=dev), torch.empty(C, device=dev)
        mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
        if training:
            nblk = max(1, min(1024, M // 64))
            part = torch.empty(nblk, 2, C, device=dev)
            K.bn_stats(x, M, C, part, nblk)
            K.bn_finalize(part, nblk, C, M, None, gamma, beta, rm, rv, scale, shift, mean, rstd, momentum=momentum, eps=eps)
        else:
            K.bn_finalize(None, 0, C, 0, None, gamma, beta, rm, rv, scale, shift, momentum=momentum, eps=eps, eval_mode=True)
        out = torch.empty_like(x)
        K.affine_act(x, M, C, scale, shift, act, out)
        ctx.save_for_backward(x, gamma, scale, shift, mean, rstd)
        ctx.cfg = (training, act, M, C)
        ctx.beta_ptr = beta.data_ptr()
        return out

    @staticmethod
    def backward(ctx, dy):
        x, gamma, scale, shift, mean, rstd = ctx.saved_tensors
        training, act, M, C = ctx.cfg
        if not training:
            raise RuntimeError("backward through an eval-mode BatchNorm is not supported (the reference only trains in train mode)")
        dy = _c(dy)
        dev = x.device
        nblk = max(1, min(1024, M // 64))
        part = torch.empty(nblk, 2, C, device=dev)
        coef = torch.empty(3, C, device=dev)
        sg, sb = _sink(gamma), GRAD_SINK.get(ctx.beta_ptr)
        K.bn_bwd_reduce(dy, None, x, M, C, scale, shift, mean, rstd, act, part, nblk)
        if sg is not None:
            assert sb is not None, "a sunk BatchNorm weight gradient needs its bias gradient sunk too"
            dgamma = dbeta = None
            K.bn_bwd_finalize(part, nblk, C, M, gamma, mean, rstd, sg, sb, coef, accumulate=True)
        else:
            dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
            K.bn_bwd_finalize(part, nblk, C, M, gamma, mean, rstd, dgamma, dbeta, coef, accumulate=False)
        dx = torch.empty_like(x)
        K.bn_bwd_apply(dy, None, x, M, C, scale, shift, act, coef, dx)
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm(x, bn, training: bool, act: Optional[str] = None):
    """bn: a BatchNormParams holder (weight, bias, running_mean, running_var, momentum, eps); act fused: 'relu' | 'mish' | None"""
    if training and K._REC is None:      # (a recorded plan's engine counts its replays: engine_functional.py flush_counters)
        bn.num_batches_tracked += 1
    return _BatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training), bn.momentum, bn.eps, act)


# ---- elementwise -----------------------------------------------------------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        _chk(x)
        x = _c(x)
        if x.numel() % 4:
            raise ValueError("activation: element count must be a multiple of 4")
        out = torch.empty_like(x)
        K.affine_act(x, x.numel() // 4, 4, None, None, act, out)
        ctx.save_for_backward(x)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        K.act_bwd(x, _c(dy), x.numel(), ctx.act, dx)
        return dx, None


def relu(x):
    return _Act.apply(x, "relu")


def mish(x):
    return _Act.apply(x, "mish")


def tanh(x):
    return _Act.apply(x, "tanh")


class _PReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        _chk(x, alpha)
        x = _c(x)
        out = torch.empty_like(x)
        K.prelu_fwd(x, alpha, x.numel(), out)
        ctx.save_for_backward(x, alpha)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        nb = 256
        dx = torch.empty_like(x)
        dap = torch.empty(nb, device=x.device)
        K.prelu_bwd(x, alpha, _c(dy), None, x.numel(), dx, dap, nb)
        da = torch.empty(1, device=x.device)
        K.reduce_partials(dap, nb, 1, da, accumulate=False)
        return dx, da


def prelu(x, alpha):
    return _PReLU.apply(x, alpha)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        if a.shape != b.shape:
            raise ValueError(f"add: {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a, memory_format=torch.contiguous_format)
        K.add(_c(a), _c(b), a.numel(), out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _Fork(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (the identity branch of a residual block): the two incoming gradients are summed by
    tpgsr_add here, not by autograd's own accumulation (an ATen kernel -- and invisible to a recorded plan)"""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        out = torch.empty_like(ga, memory_format=torch.contiguous_format)
        K.add(_c(ga), _c(gb), ga.numel(), out)
        return out


def fork(x):
    return _Fork.apply(x)


class _Cat(torch.autograd.Function):
    """torch.cat(xs, channel axis) in NHWC = interleaving channel slices"""

    @staticmethod
    def forward(ctx, *xs):
        _chk(*xs)
        lead = xs[0].shape[:-1]
        cs = [x.shape[-1] for x in xs]
        if any(x.shape[:-1] != lead for x in xs):
            raise ValueError("cat: leading dimensions differ")
        Ct = sum(cs)
        out = _new(xs[0], *lead, Ct)
        M = out.numel() // Ct
        off = 0
        for x, c in zip(xs, cs):
            K.copy_strided(_c(x), c, 0, out, Ct, off, M, c)
            off += c
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        Ct = g.shape[-1]
        M = g.numel() // Ct
        outs, off = [], 0
        for c in ctx.cs:
            d = _new(g, *g.shape[:-1], c)
            K.copy_strided(g, Ct, off, d, c, 0, M, c)
            outs.append(d)
            off += c
        return tuple(outs)


def cat(xs: Sequence[torch.Tensor]):
    return _Cat.apply(*xs)


# ---- pooling / resampling -------------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        _chk(x)
        x = _c(x)
        N, H, W, C = x.shape
        OH, OW = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        out = _new(x, N, OH, OW, C)
        K.pool2d_fwd(x, N, H, W, C, None, None, "none", k, s, p, out)
        ctx.save_for_backward(x)
        ctx.cfg = (k, s, p)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        k, s, p = ctx.cfg
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        K.pool2d_bwd(x, _c(g), N, H, W, C, None, None, "none", k, s, p, dx)
        return dx, None, None, None


def max_pool2d(x, kernel, stride=None, padding=0):
    k = (kernel, kernel) if isinstance(kernel, int) else tuple(kernel)
    s = k if stride is None else ((stride, stride) if isinstance(stride, int) else tuple(stride))
    p = (padding, padding) if isinstance(padding, int) else tuple(padding)
    return _MaxPool.apply(x, k, s, p)


class _Nearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        _chk(x)
        x = _c(x)
        N, H, W, C = x.shape
        out = _new(x, N, H * s, W * s, C)
        K.resize_nearest_fwd(x, N, H, W, C, s, out)
        ctx.cfg = (N, H, W, C, s)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C, s = ctx.cfg
        dx = _new(g, N, H, W, C)
        K.resize_nearest_bwd(_c(g), N, H, W, C, s, dx)
        return dx, None


def upsample_nearest(x, scale: int):
    """F.interpolate(x, scale_factor=scale) (mode 'nearest')"""
    return _Nearest.apply(x, int(scale))


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW):
        _chk(x)
        x = _c(x)
        N, H, W, C = x.shape
        out = _new(x, N, OH, OW, C)
        K.resize_bilinear_fwd(x, N, H, W, C, OH, OW, out)
        ctx.cfg = (N, H, W, C, OH, OW)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C, OH, OW = ctx.cfg
        dx = _new(g, N, H, W, C)
        K.resize_bilinear_bwd(_c(g), N, H, W, C, OH, OW, dx)
        return dx, None, None


def interpolate_bilinear(x, size: Tuple[int, int]):
    """F.interpolate(x, size, mode='bilinear', align_corners=True)"""
    return _Bilinear.apply(x, int(size[0]), int(size[1]))


class _HMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = _c(x)
        N, H, W, C = x.shape
        out = _new(x, N, W, C)
        K.hreduce(x, N, H, W, C, 1.0 / H, out)
        ctx.cfg = (N, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        N, H, W, C = ctx.cfg
        dx = _new(g, N, H, W, C)
        K.hbroadcast(_c(g), N, H, W, C, 1.0 / H, dx)
        return dx


def mean_over_height(x):
    """nn.AdaptiveAvgPool2d((None, 1)) applied to the (b, w, c, h) permutation of a feature map: (N, H, W, C) -> (N, W, C)"""
    return _HMean.apply(x)


# ---- bidirectional GRU over one spatial axis (GruBlock, model/tsrn.py:491-508) --------------------------------------------
class _GruProj(torch.autograd.Function):
    """gi [N][H][W][192] = x W_ih^T + b_ih for both directions (two MFMA 1x1 convs into the two column halves)"""

    @staticmethod
    def forward(ctx, x, w0, w1, b0, b1):
        _chk(x, w0, w1, b0, b1)
        x = _c(x)
        N, H, W, Cin = x.shape
        G = w0.shape[0]
        gi = _new(x, N, H, W, 2 * G)
        packs = []
        for d, (w, b) in enumerate(((w0, b0), (w1, b1))):
            wt_f, wt_d, *_ = _pack(w.reshape(G, Cin, 1, 1), False, 1.0)
            K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, Cin, G), x, wt_f, gi, bias=b, out_ld=2 * G, out_coff=d * G))
            packs.append(wt_d)
        ctx.save_for_backward(x, *packs)
        ctx.cfg = (N, H, W, Cin, G)
        return gi

    @staticmethod
    def backward(ctx, dgi):
        x, wd0, wd1 = ctx.saved_tensors
        N, H, W, Cin, G = ctx.cfg
        dgi = _c(dgi)
        outs = []
        dx = None
        for d, wt_d in enumerate((wd0, wd1)):
            g = ConvGeom(N, H, W, Cin, G)
            Z = K.wgrad_splits(g.M, g.K, G)
            part, dbp = _new(x, Z, g.K, G), _new(x, Z, G)
            K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(g, x), dgi, part, dbp, dy_ld=2 * G, dy_coff=d * G))
            dw, db = _new(x, G, Cin), _new(x, G)
            K.wgrad_reduce(part, dbp, Z, g, dw, db, accumulate=False)
            outs.append((dw, db))
            if ctx.needs_input_grad[0]:
                t = _new(x, N, H, W, Cin)
                K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, G, Cin), dgi, wt_d, t, in_ld=2 * G, in_coff=d * G))
                if dx is None:
                    dx = t
                else:
                    K.add(dx, t, dx.numel(), dx)
        return dx, outs[0][0], outs[1][0], outs[0][1], outs[1][1]


class _GruCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, whh0, whh1, bhh0, bhh1, axis):
        _chk(gi, whh0, whh1, bhh0, bhh1)
        gi = _c(gi)
        N, H, W, G2 = gi.shape
        if G2 != 192 or whh0.shape != (96, 32):
            raise NotImplementedError("the fused BiGRU kernel is specialised for hidden size 32 (the reference's hidden_units)")
        dev = gi.device
        whh = torch.empty(2, 96, 32, device=dev)
        bhh = torch.empty(2, 96, device=dev)
        for d, (w, b) in enumerate(((whh0, bhh0), (whh1, bhh1))):
            K.copy(_c(w), whh[d], 96 * 32)
            K.copy(_c(b), bhh[d], 96)
        h = _new(gi, N, H, W, 64)
        gates = _new(gi, N, H, W, 256)
        K.bigru_fwd(gi, whh, bhh, N, H, W, axis, h, gates)
        ctx.save_for_backward(gates, h, whh)
        ctx.cfg = (N, H, W, axis)
        return h

    @staticmethod
    def backward(ctx, dh):
        gates, h, whh = ctx.saved_tensors
        N, H, W, axis = ctx.cfg
        dgi, dgh = _new(h, N, H, W, 192), _new(h, N, H, W, 192)
        K.bigru_bwd(gates, h, _c(dh), None, whh, N, H, W, axis, dgi, dgh)
        res = []
        for d in range(2):
            sgn = 1 if d == 0 else -1
            # dW_hh[d] = dgh[:, d]^T h_prev(d): h shifted one step against the scan direction (engine.GruLayer.bwd)
            gh = ConvGeom(N, H, W, 32, 96, 1, 1, sgn if axis == 1 else 0, sgn if axis == 0 else 0, H, W)
            Z = K.wgrad_splits(gh.M, gh.K, 96)
            part, dbp = _new(h, Z, 32, 96), _new(h, Z, 96)
            K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(gh, h, in_ld=64, in_coff=32 * d), dgh, part, dbp, dy_ld=192, dy_coff=96 * d))
            dw, db = _new(h, 96, 32), _new(h, 96)
            K.wgrad_reduce(part, dbp, Z, gh, dw, db, accumulate=False)
            res.append((dw, db))
        return dgi, res[0][0], res[1][0], res[0][1], res[1][1], None


def bigru(x, gru, axis: int):
    """bidirectional GRU (hidden 32) along W (axis 0) or H (axis 1) of an NHWC map; gru: a GRUParams holder"""
    gi = _GruProj.apply(x, gru.weight_ih_l0, gru.weight_ih_l0_reverse, gru.bias_ih_l0, gru.bias_ih_l0_reverse)
    return _GruCore.apply(gi, gru.weight_hh_l0, gru.weight_hh_l0_reverse, gru.bias_hh_l0, gru.bias_hh_l0_reverse, axis)


# ---- STN: TPS grid + bilinear sampler (model/tps_spatial_transformer.py:97-112) --------------------------------------------
class _TpsGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ctrl, inv_kernel, coord_repr, HW):
        _chk(ctrl, inv_kernel, coord_repr)
        ctrl = _c(ctrl)
        N, NC = ctrl.shape[0], ctrl.shape[1]
        grid, src = _new(ctrl, N, HW, 2), _new(ctrl, N, HW, 2)
        K.tps_grid_fwd(ctrl, inv_kernel, coord_repr, N, HW, NC, grid, src)
        ctx.save_for_backward(src, inv_kernel, coord_repr)
        ctx.cfg = (N, HW, NC)
        ctx.mark_non_differentiable(src)
        return grid, src

    @staticmethod
    def backward(ctx, dgrid, _dsrc):
        src, inv_kernel, coord_repr = ctx.saved_tensors
        N, HW, NC = ctx.cfg
        dctrl = _new(src, N, NC, 2)
        K.tps_grid_bwd(_c(dgrid), src, inv_kernel, coord_repr, N, HW, NC, dctrl)
        return dctrl, None, None, None


class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grid, OH, OW, align_corners):
        _chk(x, grid)
        x, grid = _c(x), _c(grid)
        N, H, W, C = x.shape
        out = _new(x, N, OH, OW, C)
        K.grid_sample_fwd(x, grid, N, H, W, C, OH, OW, align_corners, out)
        ctx.save_for_backward(x, grid)
        ctx.cfg = (N, H, W, C, OH, OW, align_corners)
        return out

    @staticmethod
    def backward(ctx, g):
        x, grid = ctx.saved_tensors
        N, H, W, C, OH, OW, ac = ctx.cfg
        din = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dgrid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        K.grid_sample_bwd(x, grid, _c(g), N, H, W, C, OH, OW, ac, din, dgrid)
        return din, dgrid, None, None, None


def tps_grid(ctrl, inv_kernel, coord_repr, HW):
    return _TpsGrid.apply(ctrl, inv_kernel, coord_repr, HW)


def grid_sample(x, grid, out_hw, align_corners=False):
    return _GridSample.apply(x, grid, int(out_hw[0]), int(out_hw[1]), bool(align_corners))


# ---- evaluation-only helpers of the ASTER recognizer (model/recognizer/*) -------------------------------------------------
class PackedLinear:
    """y = x W^T + b with the weight packed (and split, under the bf16 matrix-core policies) ONCE: decode loops call the same
    nn.Linear a hundred times.  No autograd (evaluation paths only)."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor] = None):
        _chk(w, b)
        self.Cout, self.Cin = w.shape
        self.wt_f, _, _, _, _, _ = _pack(_c(w.detach()).reshape(self.Cout, self.Cin, 1, 1), False, 1.0)
        self.b = None if b is None else _c(b.detach())

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """out: optional contiguous (rows, Cout) destination (a row block of a decode loop's result)"""
        _chk(x, out)
        x = _c(x)
        rows = x.numel() // self.Cin
        if out is None:
            out = _new(x, rows, self.Cout)
        elif not out.is_contiguous() or out.numel() != rows * self.Cout:
            raise ValueError(f"PackedLinear: out must be a contiguous ({rows}, {self.Cout}) tensor")
        K.conv_fwd(K.make_conv_args(ConvGeom(1, 1, rows, self.Cin, self.Cout), x, self.wt_f, out, bias=self.b))
        return out


def bilstm_eval(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
    """one bidirectional LSTM layer, batch-first (N, T, C) -> (N, T, 2 Hh), zero initial state, evaluation only: the input
    projections of both directions as two linear launches, then per time step the split-K recurrent GEMM + the gate kernel of the
    text-prior generator (csrc/crnn.hip)"""
    _chk(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
    if x.requires_grad:
        raise RuntimeError("bilstm_eval is an evaluation path (no gradient)")
    with torch.no_grad():
        N, T, Cin = x.shape
        Hh = w_hh.shape[1]
        G4 = 4 * Hh
        xf = _c(x).reshape(N * T, Cin)
        G = cat([PackedLinear(w_ih, b_ih)(xf).reshape(N, 1, T, G4), PackedLinear(w_ih_r, b_ih_r)(xf).reshape(N, 1, T, G4)])   # [N][T][2][4Hh]
        whh = _new(x, 2, Hh, G4)
        bhh = _new(x, 2, G4)
        for d, (w, b) in enumerate(((w_hh, b_hh), (w_hh_r, b_hh_r))):
            K.pack_conv_weight(_c(w.detach()).reshape(G4, Hh, 1, 1), G4, Hh, 1, 1, whh[d], None)
            K.copy(_c(b.detach()), bhh[d], G4)
        S = max(1, Hh // 32)
        gh = _new(x, S, 2, N, G4)
        Cst, out = _new(x, N, T, 2, Hh), _new(x, N, T, 2 * Hh)
        for s in range(T):
            if s > 0:
                a = [out.data_ptr() + 4 * ((s - 1 if d == 0 else T - s) * 2 * Hh + d * Hh) for d in range(2)]
                K.lstm_rec_gemm(a[0], a[1], T * 2 * Hh, whh[0], whh[1], N, Hh, G4, S, gh)
            K.lstm_step_fwd(G, gh if s > 0 else None, S, bhh, Cst, out, N, T, Hh, s)
        return out


# ---- evaluation-only helpers of the MORAN recognizer (model/moran/*) -------------------------------------------------------
def signed_relu_pool_diff(x, kernel, stride):
    """max_pool(relu(x)) - max_pool(relu(-x)) (morn.py:61-63), x NHWC, no autograd: two fused scale-activation-pool launches and a
    subtraction"""
    _chk(x)
    x = _c(x)
    N, H, W, C = x.shape
    k = (kernel, kernel) if isinstance(kernel, int) else tuple(kernel)
    s = (stride, stride) if isinstance(stride, int) else tuple(stride)
    OH, OW = (H - k[0]) // s[0] + 1, (W - k[1]) // s[1] + 1
    pos, neg = _new(x, N, OH, OW, C), _new(x, N, OH, OW, C)
    minus, zero = torch.full((C,), -1.0, dtype=F32, device=x.device), torch.zeros(C, dtype=F32, device=x.device)
    K.pool2d_fwd(x, N, H, W, C, None, None, "relu", k, s, (0, 0), pos)
    K.pool2d_fwd(x, N, H, W, C, minus, zero, "relu", k, s, (0, 0), neg)
    K.scale_(neg, neg.numel(), minus[:1])
    out = torch.empty_like(pos)
    K.add(pos, neg, pos.numel(), out)
    return out


def offset_grid_y(grid, dy):
    """sampling grid (N, H, W, 2) of (x, y) with dy (N, H, W, 1) added to its y coordinates (morn.py:66-67), no autograd"""
    _chk(grid, dy)
    grid, dy = _c(grid), _c(dy)
    out = torch.empty_like(grid)
    M = grid.numel() // 2
    K.copy(grid, out, grid.numel())
    K.copy_strided(dy, 1, 0, out, 2, 1, M, 1, accumulate=True)
    return out
