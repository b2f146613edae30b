"""Static execution plans for the TSRN super-resolution network (forward + hand-scheduled backward).

The reference expresses the network as ~120 small nn.Modules whose forward/backward PyTorch turns into several hundred
ATen/cuDNN launches plus permute/contiguous copies (model/tsrn.py:62-78, :373-394, :491-508).  Here the whole network
is ONE recorded list of C-ABI kernel launches per (batch, height, width, mode): activations live in NHWC workspaces
allocated once, train-mode BatchNorm / mish / residual adds ride on the consumer conv's tile loader, parameter
gradients are written straight into a flat gradient arena (one RCCL all-reduce bucket, one fused Adam), and the plan
replays with no Python tensor bookkeeping -- which also makes it capturable into a hipGraph.

Module layout / parameter names stay the reference's (tpgsr_amd/model/tsrn.py), so checkpoints interchange.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import kernels as K
from ._lib import PackDesc
from .kernels import ConvGeom, Plan, recording

F32 = torch.float32


# =================================================================================================================
# flat parameter / gradient arenas
# =================================================================================================================
class ParamArena:
    """All parameters of a module as views of ONE flat fp32 buffer (and their .grad as views of a second one).
    With `external` set (ArenaPool) the two buffers are slices of a pool shared by several modules, so the whole
    job's gradients are one contiguous RCCL bucket."""

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.flat: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None
        self.offsets: Dict[str, int] = {}
        self.numel = 0
        self.external = None      # (flat slice, grad slice) handed out by an ArenaPool

    def layout(self):
        """name -> offset (floats) with every tensor 16-byte aligned; total size"""
        off, offsets = 0, {}
        for name, p in self.module.named_parameters():
            if p.dtype != F32:
                raise TypeError(f"tpgsr_amd runs fp32 parameters (got {p.dtype} for {name})")
            offsets[name] = off
            off += (p.numel() + 3) // 4 * 4
        return offsets, off

    def _walk(self):
        """(name, parameter, the `_parameters` dict that holds it, its key there) in named_parameters() order"""
        memo, out = set(), []
        for prefix, mod in self.module.named_modules():
            for k, p in mod._parameters.items():
                if p is None or p in memo:
                    continue
                memo.add(p)
                out.append((prefix + ("." if prefix else "") + k, p, mod._parameters, k))
        return out

    def _remember(self, walk):
        """the per-call check list: (holder dict, key, parameter, its address in the arena, its gradient's address, its gradient VIEW).
        The views are built once: torch's optimizers drop every .grad each step (`zero_grad()` sets them to None by default) and
        attach_grads() puts them back -- slicing and reshaping 196 views anew was 0.4 ms of every backward pass."""
        base, gbase = self.flat.data_ptr(), self.grad.data_ptr()
        self._fast = []
        for name, p, d, k in walk:
            o, n = self.offsets[name], p.numel()
            self._fast.append((d, k, p, base + 4 * o, gbase + 4 * o, self.grad[o:o + n].view(p.shape)))

    def ensure(self, device) -> bool:
        """(Re)build the arenas if the parameters are not (any more) views of them on `device`.  True if rebuilt.
        Called by every forward pass.  The common answer -- nothing moved -- costs one identity test and one data_ptr() per parameter over a
        cached list (round 6: the module walk + name lookups of the full check were 0.27 ms per call for TSRN_TL, twice per forward, a fifth
        of the drop-in loop's host time).  The cached list notices: a Parameter object replaced (`m.weight = nn.Parameter(..)`), its storage
        moved or re-typed (`.to()`, `.cuda()`, `.half()`, `p.data = ..`), the arena re-pointed by an ArenaPool.  It does not notice a
        parameter ADDED to the tree (no plan would read it)."""
        ok = self.flat is not None and self.flat.device == device
        if ok and self.external is not None:
            ok = self.flat.data_ptr() == self.external[0].data_ptr() and self.external[0].device == device
        fast = self.__dict__.get("_fast")
        if ok and fast is not None:
            for d, k, p, e, _g, _v in fast:
                if d.get(k) is not p or p.data_ptr() != e:
                    ok = False
                    break
            if ok:
                return False
        walk = self._walk()
        if ok:
            base = self.flat.data_ptr()
            for name, p, _d, _k in walk:
                if name not in self.offsets or p.data_ptr() != base + 4 * self.offsets[name] or p.dtype != F32:
                    ok = False
                    break
        if ok:
            self._remember(walk)
            return False
        self.offsets, off = self.layout()
        self.numel = off
        if self.external is not None and self.external[0].device == device:
            flat, grad = self.external
            assert flat.numel() == off and grad.numel() == off
            grad.zero_()
        else:
            self.external = None
            flat = torch.zeros(off, dtype=F32, device=device)
            grad = torch.zeros(off, dtype=F32, device=device)
        with torch.no_grad():
            for name, p, _d, _k in walk:
                o, n = self.offsets[name], p.numel()
                flat[o:o + n].copy_(p.data.reshape(-1).to(device))
                p.data = flat[o:o + n].view(p.shape)
                p.grad = grad[o:o + n].view(p.shape)
        self.flat, self.grad = flat, grad
        self._remember(walk)
        return True

    def attach_grads(self) -> bool:
        """Re-attach .grad views after an optimizer.zero_grad(set_to_none=True); the arena is zeroed in that case.
        (walks the list ensure() cached during this step's forward pass)"""
        fresh = False
        fast = self.__dict__.get("_fast")
        if fast is None:
            self._remember(self._walk())
            fast = self._fast
        for _d, _k, p, _e, ge, gv in fast:
            g = p.grad
            if g is None or g.data_ptr() != ge:
                if not fresh:
                    self.grad.zero_()
                    fresh = True
                p.grad = gv
        return fresh


class ArenaPool:
    """ONE flat parameter buffer and ONE flat gradient buffer for a list of modules (SR nets first, then the student
    recognisers): the gradient exchange of a data-parallel step is then one contiguous range per bucket
    (tpgsr_amd.distributed.GradientExchanger), zero_grad is one memset, and the modules' own ParamArena / engines keep
    working on their slices.  Slices start on 256-byte boundaries."""

    ALIGN = 64

    def __init__(self, modules):
        self.modules = []
        for m in modules:
            if all(m is not q for q in self.modules):
                self.modules.append(m)
        self.flat = self.grad = None
        self.ranges: Dict[int, tuple] = {}

    def bind(self, device):
        sizes = [m._engine().arena.layout()[1] for m in self.modules]
        if self.flat is not None and self.flat.device == device and [self.ranges[id(m)][1] - self.ranges[id(m)][0]
                                                                      for m in self.modules] == sizes:
            # fast path only while every module's arena still points into THIS pool (a module handed to a second pool -- two train
            # steps sharing a network -- was re-pointed there: rebuild rather than exchange / optimise a stale buffer)
            lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
            mine = all(m._engine().arena.external is not None and lo <= m._engine().arena.external[0].data_ptr() < hi
                       and m._engine().arena.external[0].data_ptr() == lo + 4 * self.ranges[id(m)][0] for m in self.modules)
            if mine:
                for m in self.modules:
                    m._engine().bind(device)
                return
        off, self.ranges = 0, {}
        for m, n in zip(self.modules, sizes):
            self.ranges[id(m)] = (off, off + n)
            off = (off + n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(off, dtype=F32, device=device)
        self.grad = torch.zeros(off, dtype=F32, device=device)
        for m in self.modules:
            a, b = self.ranges[id(m)]
            eng = m._engine()
            eng.arena.external = (self.flat[a:b], self.grad[a:b])
            eng.device = None          # force a re-bind against the pooled storage
            eng.bind(device)

    def span(self, modules):
        """[begin, end) in floats covering the given (adjacent) modules"""
        r = [self.ranges[id(m)] for m in modules]
        return min(a for a, _ in r), max(b for _, b in r)


class _Ws:
    """Named workspace tensors (allocated once per plan set)."""

    def __init__(self, device):
        self.device = device
        self.t: Dict[str, torch.Tensor] = {}

    def __call__(self, name, *shape, dtype=F32):
        t = self.t.get(name)
        if t is None:
            t = torch.empty(*shape, dtype=dtype, device=self.device)
            self.t[name] = t
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), shape)
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.t.values() if isinstance(t, torch.Tensor))


# =================================================================================================================
# layers: parameters + packed operands + launch helpers
# =================================================================================================================
class ConvLayer:
    def __init__(self, eng, wname: str, bname: Optional[str], KH=1, KW=1, pad_h=0, pad_w=0, wscale=1.0, tail=False,
                 need_dgrad=True):
        self.eng, self.wname, self.bname, self.tail, self.wscale = eng, wname, bname, tail, wscale
        self.w = eng.P[wname]
        self.b = eng.P[bname] if bname else None
        dev = eng.device
        if tail:  # folded 9x9 C->Co conv: a 9x1 conv with KS*Co output columns
            Co, Cc, KS, _ = self.w.shape
            self.Co, self.KS = Co, KS
            self.Cout, self.Cin, self.KH, self.KW, self.pad_h, self.pad_w = KS * Co, Cc, KS, 1, KS // 2, 0
        else:
            self.Cout, self.Cin = self.w.shape[0], self.w.shape[1]
            self.KH, self.KW, self.pad_h, self.pad_w = KH, KW, pad_h, pad_w
        Kd = self.KH * self.KW * self.Cin
        self.wt_f = torch.empty(Kd, self.Cout, dtype=F32, device=dev)
        self.wt_d = torch.empty(self.KH * self.KW * self.Cout, self.Cin, dtype=F32, device=dev) if need_dgrad else None
        if tail:
            eng.add_pack(self.w, self.wt_f, self.wt_d, Cout=self.Co, Cin=self.Cin, KH=self.KS, KW=self.KS, kind=1)
        else:
            eng.add_pack(self.w, self.wt_f, self.wt_d, Cout=self.Cout, Cin=self.Cin, KH=KH, KW=KW, kind=0,
                         f_ld=self.Cout, wscale=wscale)
        eng.register_operand(self.wt_f, self.wt_d, cins=(self.Cin, self.Cout))

    def geom(self, N, H, W) -> ConvGeom:
        return ConvGeom(N, H, W, self.Cin, self.Cout, self.KH, self.KW, self.pad_h, self.pad_w)

    def fwd(self, N, H, W, x, out, *, use_bias=True, **kw) -> ConvGeom:
        g = self.geom(N, H, W)
        a = K.make_conv_args(g, x, self.wt_f, out, bias=self.b if (use_bias and not self.tail) else None, **kw)
        K.conv_fwd(a)
        g.bn_row_tiles = max(1, a.bn_row_tiles)       # granularity of the statistics rows this launch leaves (kernels.make_conv_args)
        return g

    def dgrad(self, N, H, W, dy, dx, **kw):
        """dx[N][H][W][Cin] = conv(dy[N][OH][OW][Cout], wt_d)"""
        K.conv_fwd(K.make_conv_args(self.geom(N, H, W).dgrad(), dy, self.wt_d, dx, **kw))

    def dgrad_takes_folded_apply(self, N, H, W, dz, y, coef) -> bool:
        """can this layer's data gradient take dy = c0 dz + c1 y + c2 (a BatchNorm's backward apply) through its LOADER -- i.e. does the
        launch land on the whole-CU halo kernel, whose loader has the scaled residual operand (tpgsr_conv_args.in2_scale)?"""
        if not (K.BNB_APPLY_FOLD and K.CONV_TERMS and not K.DRYRUN_NO_LIB()):
            return False
        a = K.make_conv_args(self.geom(N, H, W).dgrad(), dz, self.wt_d, dz, in_scale=coef[0], in_shift=coef[2], in2=y, in2_scale=coef[1])
        return K.conv_in2_scale_ok(a)

    def wgrad(self, N, H, W, x, dy, *, loader: dict = None, dy_kw: dict = None):
        """dW += A^T dy and db += colsum(dy), straight into the gradient arena."""
        eng = self.eng
        g = self.geom(N, H, W)
        Z = K.wgrad_splits(g.M, g.K, g.Cout, geom=g)
        has_b = self.b is not None and not self.tail
        part, dbp = eng.wgrad_buffers(Z * g.K * g.Cout, Z * g.Cout if has_b else 0)
        ca = K.make_conv_args(g, x, **(loader or {}))
        with K.side():
            K.conv_wgrad(K.make_wgrad_args(ca, dy, part, dbp, zsplits=Z, **(dy_kw or {})))
            K.wgrad_reduce(part, dbp, Z, g, eng.G[self.wname], eng.G[self.bname] if has_b else None,
                           layout=2 if self.tail else 0, accumulate=True, gscale=self.wscale)


class FoldedDgrad:
    """Data gradient of a KS x KS convolution with FEW input channels (block1, model/tsrn.py:28: 4 -> 64 over 9 x 9 taps; the STN head's
    gradient enters through it, model/tsrn.py:183-186).  As an implicit GEMM it has 4 output columns in a 32-column MFMA block (115 us at
    2 % of its peak); folded like the tail convolution -- a KS x 1 convolution over dy's channels with the KS kw-taps in the columns
    (36 of them), then tpgsr_shiftsum_nhwc adds the column groups back up -- it runs on the whole-CU halo kernel like a trunk layer."""

    def __init__(self, eng, wname: str):
        w = eng.P[wname]
        self.Cout, self.Ci, self.KS = w.shape[0], w.shape[1], w.shape[2]
        assert w.shape[2] == w.shape[3] and self.KS % 2 == 1      # (any channel count: the shift-sum stages rows element by element when KS Ci % 4 != 0)
        self.NP = self.KS * self.Ci
        self.wt = torch.empty(self.KS * self.Cout, self.NP, dtype=F32, device=eng.device)
        eng.add_pack(w, self.wt, None, Cout=self.Cout, Cin=self.Ci, KH=self.KS, KW=self.KS, kind=7)
        eng.register_operand(self.wt, cins=(self.Cout,))

    def run(self, N, H, W, dy, P, dx):
        """dx [N][H][W][Ci] from dy [N][H][W][Cout]; P: [N H W][KS Ci] scratch"""
        g = ConvGeom(N, H, W, self.Cout, self.NP, self.KS, 1, self.KS // 2, 0)
        K.conv_fwd(K.make_conv_args(g, dy, self.wt, P))
        K.shiftsum_nhwc(P, N, H, W, self.Ci, self.KS, dx)


class BNLayer:
    def __init__(self, eng, prefix: str, pad_to: int = 0):
        self.eng, self.prefix = eng, prefix
        self.gamma, self.beta = eng.P[prefix + ".weight"], eng.P[prefix + ".bias"]
        self.rm, self.rv = eng.B[prefix + ".running_mean"], eng.B[prefix + ".running_var"]
        self.C = C_ = self.gamma.numel()
        self.pad_to = max(C_, pad_to)
        eng._bn_layers.append(self)
        self.scale = self.shift = self.save_mean = self.save_rstd = self.coef = None   # bound per plan by use(ws)
        # ticket counters of the convolution launches that finalize this BatchNorm themselves (forward / backward; left at zero by the
        # last workgroup; launches sharing one never overlap: they are the same layer's, in stream order)
        self.tickets = torch.zeros(2, dtype=torch.int32, device=eng.device)

    def use(self, ws):
        """Batch statistics / folded scale+shift belong to ONE forward: they live in the plan's workspace (a network
        shared between cascade stages runs several forwards before the first backward)."""
        if ("bn", self.prefix) not in ws.t:
            dev, C_ = self.eng.device, self.C
            # pad_to > C: identity (1, 0) tail so a concatenated loader can index scale/shift past this BN's channels
            ws.t[("bn", self.prefix)] = (torch.ones(self.pad_to, dtype=F32, device=dev), torch.zeros(self.pad_to, dtype=F32, device=dev),
                                         torch.empty(C_, dtype=F32, device=dev), torch.empty(C_, dtype=F32, device=dev),
                                         torch.empty(3, C_, dtype=F32, device=dev))
        self.scale, self.shift, self.save_mean, self.save_rstd, self.coef = ws.t[("bn", self.prefix)]

    def partial(self, M):
        """scratch for the producing conv's epilogue statistics"""
        nblk = (M + 63) // 64
        return self.eng.scratch("bn_partial" + K.stream_tag(), nblk * 2 * self.C), nblk

    def finalize(self, M, conv_bias, training, row_tiles=1):
        if training:
            part, _ = self.partial(M)
            nblk = K.bn_rows(M, row_tiles)
            K.bn_finalize(part, nblk, self.C, M, conv_bias, self.gamma, self.beta, self.rm, self.rv, self.scale, self.shift,
                          self.save_mean, self.save_rstd)
        else:
            K.bn_finalize(None, 0, self.C, 0, None, self.gamma, self.beta, self.rm, self.rv, self.scale, self.shift,
                          eval_mode=True)

    def derive_fwd(self, M, conv_bias, row_tiles=1):
        """descriptor (kernels.make_bn_derive) for the FIRST consumer of this training-mode BatchNorm's output: that launch reduces the
        producing convolution's partial rows itself (csrc/bn_derive.h) and finalize() is NOT recorded; None when the channel count is
        outside the prologue's range or the switch is off (then: finalize() + the plain consumer)"""
        if not K.bn_derive_ok(self.C, M):
            return None
        part, _ = self.partial(M)
        return K.make_bn_derive(part, K.bn_rows(M, row_tiles), self.C, M, self.gamma, bias=conv_bias, beta=self.beta, running_mean=self.rm,
                                running_var=self.rv, scale=self.scale, shift=self.shift, save_mean=self.save_mean,
                                save_rstd=self.save_rstd)

    def fin(self, M, conv_bias):
        """kwargs (`bn_fin=`) for the training-mode convolution that leaves this BatchNorm's statistics in partial(M): the launch
        finalizes them itself (then finalize() is NOT recorded), or None when that is switched off / the fp32 kernels run"""
        if not K.bn_fin_fused():
            return None
        return dict(mode=1, count=M, counter=self.tickets[0:1], gamma=self.gamma, beta=self.beta, bias=conv_bias, scale=self.scale,
                    shift=self.shift, save_mean=self.save_mean, save_rstd=self.save_rstd, running_mean=self.rm, running_var=self.rv)

    @property
    def loader(self):
        return dict(in_scale=self.scale, in_shift=self.shift)

    def fuse_stats(self, y, M, act, cin, store_dz=False):
        """kwargs (`bnb=`) for the convolution over `cin` channels that PRODUCES this BatchNorm's incoming gradient: its epilogue then
        leaves the reduction sums of the backward pass behind and backward(..., fused=True) skips the statistics launch.  None when
        the launch would not run on a kernel that has the epilogue (fp32 policy, switch off)."""
        if not K.bnb_fusable(cin):
            return None
        nblk = (M + 63) // 64
        part = self.eng.scratch("bnb_partial" + K.stream_tag(), nblk * 2 * self.C)
        d = dict(y=y, mean=self.save_mean, rstd=self.save_rstd, scale=self.scale, shift=self.shift, act=act, partial=part,
                 store_dz=bool(store_dz),          # the producer stores dz = da act'(.) instead of da (what backward_folded consumes)
                 coarse=not K.bn_fin_fused())      # (make_conv_args writes the granularity it settled on back as d["row_tiles"])
        if K.bn_fin_fused():       # the producing launch also reduces the sums (backward(..., fused=) then records the apply only)
            d["fin"] = dict(mode=2, count=M, counter=self.tickets[1:2], gamma=self.gamma, coef=self.coef,
                            dgamma=self.eng.G[self.prefix + ".weight"], dbeta=self.eng.G[self.prefix + ".bias"], accumulate=True)
        return d

    def backward_folded(self, dz, y, M, dy, fused):
        """The backward pass with its apply launch OFF the caller's stream (round 6).  `dz` = the incoming gradient
        with the activation's derivative already in it (the producer's epilogue: fuse_stats(..., store_dz=True), or no activation), its
        statistics in fused["partial"].  Records the finalize (coefficients, dgamma / dbeta) on the caller's stream, the apply
        dy = c0 dz + c1 y + c2 on the WEIGHT-GRADIENT stream (the weight gradient that follows there is its only reader), and returns the
        loader kwargs with which the consuming data-gradient convolution reads `dz` as if it were dy.  `dz` must not be recycled before the
        weight-gradient stream is joined (the engines give it a buffer of its own)."""
        eng = self.eng
        assert fused is not None and fused["y"] is y and (fused["act"] in (None, "none") or fused.get("store_dz"))
        nblk, part = K.bn_rows(M, fused.get("row_tiles", 1)), fused["partial"]
        K.bn_bwd_finalize(part, nblk, self.C, M, self.gamma, self.save_mean, self.save_rstd, eng.G[self.prefix + ".weight"],
                          eng.G[self.prefix + ".bias"], self.coef, accumulate=True)
        with K.side():
            K.bn_bwd_apply(dz, None, y, M, self.C, self.scale, self.shift, "none", self.coef, dy)
        return dict(in_scale=self.coef[0], in_shift=self.coef[2], in2=y, in2_scale=self.coef[1])

    def can_fold(self, M):
        """the folded form replaces finalize + apply as recorded by backward(): not with the experimental in-launch finalizes"""
        return K.BNB_APPLY_FOLD and not K.BN_FIN_FUSE and not K.bn_derive_ok(self.C, M) and self.C % 4 == 0

    def backward(self, da, da2, y, M, act, dy, fused=None):
        """dy = dL/d(pre-BN y) from da (+da2) = dL/d act(BN(y)); accumulates dgamma/dbeta into the arena.
        fused: what fuse_stats(...) returned for the producer of `da` (same stream, no other BatchNorm backward in between)."""
        eng = self.eng
        if fused is not None:
            assert da2 is None and fused["y"] is y and fused["act"] == act
            nblk, part = K.bn_rows(M, fused.get("row_tiles", 1)), fused["partial"]
            if fused.get("fin") is not None and K.BN_FIN_FUSE:      # finalized by the producing launch
                K.bn_bwd_apply(da, da2, y, M, self.C, self.scale, self.shift, act, self.coef, dy)
                return
        else:
            nblk = min(1024, max(1, M // 64))
            part = eng.scratch("bnb_partial" + K.stream_tag(), nblk * 2 * self.C)
            K.bn_bwd_reduce(da, da2, y, M, self.C, self.scale, self.shift, self.save_mean, self.save_rstd, act, part, nblk)
        if K.bn_derive_ok(self.C, M):      # coefficients + dgamma / dbeta derived by the apply launch itself: one launch instead of two
            d = K.make_bn_derive(part, nblk, self.C, M, self.gamma, save_mean=self.save_mean, save_rstd=self.save_rstd,
                                 dgamma=eng.G[self.prefix + ".weight"], dbeta=eng.G[self.prefix + ".bias"], coef=self.coef, accumulate=True)
            K.bn_bwd_apply_bnd(d, da, da2, y, M, self.scale, self.shift, act, dy)
            return
        K.bn_bwd_finalize(part, nblk, self.C, M, self.gamma, self.save_mean, self.save_rstd, eng.G[self.prefix + ".weight"],
                          eng.G[self.prefix + ".bias"], self.coef, accumulate=True)
        K.bn_bwd_apply(da, da2, y, M, self.C, self.scale, self.shift, act, self.coef, dy)


class GruLayer:
    """GruBlock (model/tsrn.py:491-508): 1x1 conv -> bidirectional GRU(hidden 32) along one spatial axis.
    axis 0: sequences along W (gru2); axis 1: along H (gru1, the reference's transpose(-1,-2)).

    The 1x1 conv and the GRU's input projection are two linear maps back to back, so they run as ONE MFMA conv with the
    composed operand Wc = W_ih W_1, bc = W_ih b_1 + b_ih (rebuilt from the parameters every step by the pack program):
    the intermediate 64-channel map of the reference never exists, in either direction.  The gradients of W_1, b_1,
    W_ih, b_ih follow from dWc / dbc by the chain rule (tpgsr_compose_bwd_program, once per backward pass)."""

    def __init__(self, eng, prefix: str, axis: int):
        self.eng, self.prefix, self.axis = eng, prefix, axis
        P, dev = eng.P, eng.device
        self.w1name, self.b1name = prefix + ".conv1.weight", prefix + ".conv1.bias"
        W1 = P[self.w1name]
        self.U, self.Cin = W1.shape[0], W1.shape[1]
        gp = prefix + ".gru."
        self.gp = gp
        hid = P[gp + "weight_hh_l0"].shape[1]
        if hid != 32:
            raise NotImplementedError("the fused BiGRU kernel is specialised for hidden_units=32 (the reference default)")
        assert P[gp + "weight_ih_l0"].shape[1] == self.U and self.U % 4 == 0 and self.Cin % 4 == 0
        Cin = self.Cin
        self.wc_f = torch.empty(Cin, 192, dtype=F32, device=dev)    # forward operand [K=Cin][192]
        self.wc_d = torch.empty(192, Cin, dtype=F32, device=dev)    # dgrad operand  [K'=192][Cin]
        self.bc = torch.empty(192, dtype=F32, device=dev)
        self.whh = torch.empty(2, 96, 32, dtype=F32, device=dev)
        self.bhh = torch.empty(2, 96, dtype=F32, device=dev)
        for d, suf in enumerate(("", "_reverse")):
            wih = P[gp + "weight_ih_l0" + suf]
            eng.add_pack(wih, self.wc_f, self.wc_d, Cout=96, Cin=Cin, KH=self.U, KW=1, kind=5, f_ld=192, f_coff=d * 96,
                         src2=W1, numel=96 * Cin)
            eng.add_pack(wih, self.bc, None, Cout=96, Cin=0, KH=self.U, KW=1, kind=6, f_coff=d * 96, src2=P[self.b1name],
                         src3=P[gp + "bias_ih_l0" + suf], numel=96)
            eng.add_pack(P[gp + "weight_hh_l0" + suf], self.whh[d], None, kind=2)
            eng.add_pack(P[gp + "bias_hh_l0" + suf], self.bhh[d], None, kind=2)
        eng.register_operand(self.wc_f, self.wc_d)

    def fwd(self, N, H, W, x, gi, h, gates=None, **loader):
        """gi = loader(x) Wc^T + bc; h = BiGRU(gi) (gates: saved for bwd in training plans).  ONE launch (csrc/gru_proj.hip: the
        projection goes from the matrix cores into LDS, `gi` is not touched) when the block is that kernel's, else projection + scan;
        gi: the [P][192] workspace or a callable returning it (only called when it is needed)"""
        g = ConvGeom(N, H, W, self.Cin, 192)
        pa = K.make_bigru_proj_args(K.make_conv_args(g, x, self.wc_f, None, bias=self.bc, **loader), self.whh, self.bhh, self.axis, h, gates)
        if K.bigru_proj_supported(pa):
            K.bigru_proj_fwd(pa)
            return
        gi = gi() if callable(gi) else gi
        K.conv_fwd(K.make_conv_args(g, x, self.wc_f, gi, bias=self.bc, **loader))
        K.bigru_fwd(gi, self.whh, self.bhh, N, H, W, self.axis, h, gates)

    def bwd(self, N, H, W, x, gates, h, dh, dh2, dgi, dgh, dx, dx_bnb=None, **loader):
        """all parameter gradients of the block + dx = dL/d loader(x) (dx None: the caller takes it from dgi / wc_d);
        dx_bnb: BNLayer.fuse_stats(...) of the BatchNorm dx is the incoming gradient of"""
        eng, G, gp = self.eng, self.eng.G, self.gp
        if K.gru_wgrad_fused():
            # ONE weight-gradient launch for the whole block (csrc/gru_wgrad.hip): back-propagation through time writes dgi and only the
            # n-gate plane of the hidden-side gradient (`dgh` is used as [P][64]); loader(x), h, dgi, dghn are each read once
            P = N * H * W
            dghn = dgh.view(-1)[:P * 64].view(P, 64)
            K.bigru_bwd2(gates, h, dh, dh2, self.whh, N, H, W, self.axis, dgi, dghn)
            with K.side():
                Z = K.gru_wgrad_splits(P)
                nC, nH = Z * self.Cin * 192, 2 * Z * 32 * 96
                part, dbp = eng.wgrad_buffers(nC + nH, Z * 192 + 2 * Z * 96)       # one buffer each: slabs of dWc | dWhh, of dbc | dbhh
                partC, partH, dbC, dbH = part[:nC], part[nC:nC + nH], dbp[:Z * 192], dbp[Z * 192:Z * 192 + 2 * Z * 96]
                gc = ConvGeom(N, H, W, self.Cin, 192)
                K.gru_wgrad(K.make_conv_args(gc, x, **loader), dgi, dghn, h, self.axis, Z, partC, dbC, partH, dbH)
                for d, suf in enumerate(("", "_reverse")):
                    gh = ConvGeom(N, H, W, 32, 96)
                    K.wgrad_reduce(partH[d * Z * 32 * 96:(d + 1) * Z * 32 * 96], dbH[d * Z * 96:(d + 1) * Z * 96], Z, gh,
                                   G[gp + "weight_hh_l0" + suf], G[gp + "bias_hh_l0" + suf], accumulate=True)
                ws = eng._cur_ws
                dWc, dbc = ws("dWc_" + self.prefix, 192, self.Cin), ws("dbc_" + self.prefix, 192)
                K.wgrad_reduce(partC, dbC, Z, gc, dWc, dbc, accumulate=False)
                eng._compose.append((self, dWc, dbc))
            if dx is not None:
                K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, 192, self.Cin), dgi, self.wc_d, dx, bnb=dx_bnb))
            return
        K.bigru_bwd(gates, h, dh, dh2, self.whh, N, H, W, self.axis, dgi, dgh)
        with K.side():
            for d, suf in enumerate(("", "_reverse")):
                sgn = 1 if d == 0 else -1
                # hidden side: dW_hh[d] = dgh[:, d]^T h_prev(d), h_prev = h shifted one step against the scan direction
                gh = ConvGeom(N, H, W, 32, 96, 1, 1, sgn if self.axis == 1 else 0, sgn if self.axis == 0 else 0, H, W)
                Z = K.wgrad_splits(gh.M, gh.K, 96)
                part, dbp = eng.wgrad_buffers(Z * 32 * 96, Z * 96)
                ca = K.make_conv_args(gh, h, in_ld=64, in_coff=32 * d)
                K.conv_wgrad(K.make_wgrad_args(ca, dgh, part, dbp, dy_ld=192, dy_coff=96 * d))
                K.wgrad_reduce(part, dbp, Z, gh, G[gp + "weight_hh_l0" + suf], G[gp + "bias_hh_l0" + suf], accumulate=True)
            # input side, both directions at once: dWc = dgi^T loader(x), dbc = colsum(dgi); chain rule at the end of the pass
            gc = ConvGeom(N, H, W, self.Cin, 192)
            Z = K.wgrad_splits(gc.M, gc.K, 192)
            part, dbp = eng.wgrad_buffers(Z * self.Cin * 192, Z * 192)
            K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(gc, x, **loader), dgi, part, dbp))
            ws = eng._cur_ws
            dWc, dbc = ws("dWc_" + self.prefix, 192, self.Cin), ws("dbc_" + self.prefix, 192)
            K.wgrad_reduce(part, dbp, Z, gc, dWc, dbc, accumulate=False)
            eng._compose.append((self, dWc, dbc))
        if dx is not None:
            K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, 192, self.Cin), dgi, self.wc_d, dx, bnb=dx_bnb))


class TConvStrip:
    """ConvTranspose2d(Cin, Cout, 3, stride (s_h, s_w), padding (1, p_w), bias=False) on an H=1 strip (InfoGen,
    model/tsrn.py:81-108): only the kh=1 kernel row meets data, so it is a 1-D transposed conv along W, run as a
    stride-1 1x3 conv over the zero-dilated strip (forward / weight gradient) and as a stride-s_w conv over dy
    (data gradient).  An input channel count that is not a multiple of 4 (tconv1: the 37 classes) is zero-padded to
    Cp channels in the operands; the caller then hands in the padded strip [N][W][Cp]."""

    def __init__(self, eng, wname: str, stride_w: int, pad_w: int):
        self.eng, self.wname, self.sw, self.pw = eng, wname, stride_w, pad_w
        self.w = eng.P[wname]
        self.Cin, self.Cout = self.w.shape[0], self.w.shape[1]
        self.Cp = (self.Cin + 3) // 4 * 4
        assert tuple(self.w.shape[2:]) == (3, 3)
        dev = eng.device
        self.wt_f = torch.zeros(3 * self.Cp, self.Cout, dtype=F32, device=dev)
        self.wt_d = torch.zeros(3 * self.Cout, self.Cp, dtype=F32, device=dev)
        eng.add_pack(self.w, self.wt_f, self.wt_d, Cout=self.Cout, Cin=self.Cin, KH=3, KW=3, kind=4, f_ld=self.Cout,
                     d_ld=self.Cp, cin_ld=self.Cp)
        eng.register_operand(self.wt_f, self.wt_d)

    def out_w(self, Win):
        return (Win - 1) * self.sw - 2 * self.pw + 3

    def geom(self, N, Win) -> ConvGeom:
        return ConvGeom(N, 1, (Win - 1) * self.sw + 1, self.Cp, self.Cout, 1, 3, 0, 2 - self.pw)

    def fwd(self, N, Win, x, out, **kw):
        g = self.geom(N, Win)
        assert g.OW == self.out_w(Win)
        K.conv_fwd(K.make_conv_args(g, x, self.wt_f, out, in_dil_w=self.sw, **kw))
        return g

    def wgrad(self, N, Win, x, dy, loader=None):
        eng, g = self.eng, self.geom(N, Win)
        Z = K.wgrad_splits(g.M, g.K, g.Cout)
        part, _ = eng.wgrad_buffers(Z * g.K * g.Cout)
        ca = K.make_conv_args(g, x, in_dil_w=self.sw, **(loader or {}))
        with K.side():
            K.conv_wgrad(K.make_wgrad_args(ca, dy, part, None))
            K.wgrad_reduce(part, None, Z, g, eng.G[self.wname], None, layout=3, accumulate=True,
                           real=(self.Cin, 1, 3, self.Cp) if self.Cp != self.Cin else None)

    def dgrad(self, N, Win, dy, dx, bnb=None):
        """dx[N][1][Win][Cin] = strided conv of dy[N][1][OW][Cout] with the un-flipped taps"""
        g = ConvGeom(N, 1, self.out_w(Win), self.Cout, self.Cin, 1, 3, 0, self.pw, 1, Win)
        K.conv_fwd(K.make_conv_args(g, dy, self.wt_d, dx, stride_w=self.sw, wt_ld=self.Cp, bnb=bnb))


# =================================================================================================================
# TSRN engine
# =================================================================================================================
class _EngineBase:
    """Arenas, packed-operand table, stream-ordered scratch and plan cache shared by the network engines."""
    FUSED = True         # one fused autograd node per network (see tpgsr_amd.distributed.DataParallel)

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.arena = ParamArena(module)
        self.device = None
        self._plans: Dict[tuple, dict] = {}
        self._scratch: Dict[str, torch.Tensor] = {}
        self._pack: List[tuple] = []
        self._operands: List[torch.Tensor] = []
        self._split_n = 0
        self._bn_layers: List["BNLayer"] = []
        self._pending_batches = 0
        self._wg_idx, self._cur_ws = 0, None
        self._compose: List[tuple] = []
        self._live: Dict[int, int] = {}      # module-API workspace slot -> generation of the forward that owns it
        self._gen = 0
        self._kernel_writes = 0              # bumped by whoever rewrites the parameter arena through a raw pointer (FusedAdam)
        self._packed_key = None              # (arena version, kernel writes) the packed / split operands were built from
        self._packed_split = False           # ... and whether that pack rebuilt the bf16 twins too

    # ---- workspace slots of the nn.Module API --------------------------------------------------------------------
    # A training-mode forward saves its activations / BN statistics / GRU-LSTM gates in a workspace slot until its
    # backward ran.  The train-step drivers name their slots explicitly (stage index); autograd-driven callers may run
    # several forwards of one module before the first backward (fwd, fwd, bwd, bwd; a cascade with --sr_share), so every
    # such forward takes a free slot and its backward releases it.  More than MAX_LIVE outstanding forwards recycle the
    # oldest slot; a backward whose slot was recycled raises instead of using overwritten activations.
    MAX_LIVE = 8
    SLOT_BASE = 1000

    def acquire_slot(self):
        self._gen += 1
        free = [s for s in range(self.MAX_LIVE) if s not in self._live]
        s = free[0] if free else min(self._live, key=self._live.get)
        self._live[s] = self._gen
        return self.SLOT_BASE + s, self._gen

    def check_slot(self, slot, gen):
        if self._live.get(slot - self.SLOT_BASE) != gen:
            raise RuntimeError(
                f"{type(self.module).__name__}: the activations saved by this forward were overwritten (more than "
                f"{self.MAX_LIVE} training-mode forwards without a backward, or the module was re-bound in between)")

    def release_slot(self, slot, gen):
        if self._live.get(slot - self.SLOT_BASE) == gen:
            del self._live[slot - self.SLOT_BASE]

    # ---- scratch buffers live only between consecutive launches (stream-ordered reuse) ----------------------
    def scratch(self, name, numel):
        t = self._scratch.get(name)
        if t is None or t.numel() < numel:
            if K._REC is not None and getattr(K._REC, "final", False):
                raise RuntimeError(f"scratch '{name}' would be re-allocated while recording the final plan")
            t = torch.empty(max(int(numel), 1), dtype=F32, device=self.device)
            self._scratch[name] = t
        return t

    def wgrad_buffers(self, n_part, n_db=0):
        """Slab buffers of one weight-gradient launch: stream-ordered scratch when its reduce follows immediately, the
        layer's own workspace when the plan batches all reduces into one launch at the end (K.flush_wgrad_reduces)."""
        if K.deferring():
            i = self._wg_idx
            self._wg_idx += 1
            ws = self._cur_ws
            return ws(f"wgp{i}", n_part), (ws(f"wgb{i}", n_db) if n_db else None)
        return self.scratch("wgrad_part", n_part), (self.scratch("wgrad_dbpart", n_db) if n_db else None)

    def flush_compose_bwd(self):
        """ONE launch: chain rule from every GruBlock's composed-operand gradient to its conv1 / weight_ih gradients."""
        items, self._compose = self._compose, []
        if not items:
            return
        from ._lib import ComposeBwdDesc, load
        lib = load()
        arr = (ComposeBwdDesc * len(items))()
        blk, keep = 0, []
        for d, (L, dWc, dbc) in zip(arr, items):
            P, G, gp = self.P, self.G, L.gp
            t = dict(dWc=dWc, dbc=dbc, W1=P[L.w1name], b1=P[L.b1name], wih0=P[gp + "weight_ih_l0"], wih1=P[gp + "weight_ih_l0_reverse"],
                     dW1=G[L.w1name], db1=G[L.b1name], dwih0=G[gp + "weight_ih_l0"], dwih1=G[gp + "weight_ih_l0_reverse"],
                     dbih0=G[gp + "bias_ih_l0"], dbih1=G[gp + "bias_ih_l0_reverse"])
            for k, v in t.items():
                setattr(d, k, v.data_ptr())
            keep += list(t.values())
            d.Cin, d.U, d.G, d.blk0 = L.Cin, L.U, 96, blk
            blk += lib.tpgsr_compose_bwd_blocks(L.Cin, L.U, 96)
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        if K._REC is not None:
            K._REC.keep += keep
        with K.side():
            K.compose_bwd_program(table, len(items), blk)

    def add_pack(self, src, dst_f, dst_d, Cout=0, Cin=0, KH=1, KW=1, kind=0, f_ld=0, f_coff=0, wscale=1.0, src2=None, src3=None,
                 numel=None, d_ld=0, cin_ld=0):
        self._pack.append((src, dst_f, dst_d, Cout, Cin, KH, KW, kind, f_ld, f_coff, wscale, src2, src3, numel, d_ld, cin_ld,
                           getattr(self, "_pack_group", 0)))

    def register_operand(self, *tensors, cins=None):
        """fp32 MFMA operands [K][ld] packed by the pack program: get a bf16 split twin when the bf16 matrix-core path is on.
        cins: per tensor, the input-channel count of the convolution consuming it (multi-tap operands over a multiple of 32
        channels are split in channel-block order, which is what the halo kernel reads)"""
        for i, t in enumerate(tensors):
            if t is not None:
                assert t.dim() == 2 and t.is_contiguous()
                self._operands.append(t)
                t._tpgsr_cin = K.block_order_cin(t.shape[0], cins[i] if cins else 0)
                t._tpgsr_group = getattr(self, "_pack_group", 0)

    # Pack / split tables per GROUP (engine attribute `_pack_group` while the layers are built; 0 unless a network says otherwise): the
    # text-prior generator packs its first three convolutions' operands on the caller's stream and everything behind them -- 96 % of its
    # parameters -- on another stream while those convolutions run (CRNNEngine; pack_group()).
    def _finish_split_table(self):
        from ._lib import SplitDesc, load
        self._split_tabs = {}
        self._split_n = 0
        ops, seen = [], set()
        for t in self._operands:
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                ops.append(t)
        if K.POLICY == "f32" or not ops:
            return
        lib = load()
        self._split_keep = []
        for grp in sorted({getattr(t, "_tpgsr_group", 0) for t in ops}):
            sel = [t for t in ops if getattr(t, "_tpgsr_group", 0) == grp]
            arr = (SplitDesc * len(sel))()
            blk = 0
            for d, t in zip(arr, sel):
                Kd, N = t.shape
                kp = (Kd + 31) // 32 * 32
                twin = torch.zeros(3 * ((N + 31) // 32 * 32) * kp, dtype=torch.bfloat16, device=self.device)
                cin = getattr(t, "_tpgsr_cin", 0)
                K.register_bf_twin(t, twin, kp, cin)
                d.src, d.dst, d.K, d.N, d.ld, d.kp, d.blk0, d.cin = t.data_ptr(), twin.data_ptr(), Kd, N, N, kp, blk, cin
                blk += lib.tpgsr_split_bf_blocks(Kd, N)
                self._split_keep += [t, twin]
            self._split_tabs[grp] = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(sel), blk)
        self._split_n = len(ops)

    def _finish_pack_table(self):
        from ._lib import load
        lib = load()
        self._pack_tabs = {}
        self._pack_keep = []
        for grp in sorted({e[16] for e in self._pack}):
            sel = [e for e in self._pack if e[16] == grp]
            arr = (PackDesc * len(sel))()
            blk = 0
            for d, (src, dst_f, dst_d, Cout, Cin, KH, KW, kind, f_ld, f_coff, wscale, src2, src3, numel, d_ld, cin_ld, _g) in zip(arr, sel):
                d.src, d.dst_f = src.data_ptr(), dst_f.data_ptr()
                d.dst_d = dst_d.data_ptr() if dst_d is not None else None
                d.src2 = src2.data_ptr() if src2 is not None else None
                d.src3 = src3.data_ptr() if src3 is not None else None
                d.Cout, d.Cin, d.KH, d.KW, d.kind, d.f_ld, d.f_coff, d.wscale = Cout, Cin, KH, KW, kind, f_ld, f_coff, wscale
                d.d_ld, d.cin_ld = d_ld, cin_ld
                cnt = src.numel() if numel is None else numel
                d.numel, d.blk0 = cnt, blk
                blk += lib.tpgsr_pack_blocks(kind, Cout, Cin, KH, KW, cnt)
                self._pack_keep += [src, dst_f, dst_d, src2, src3]
            self._pack_tabs[grp] = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(sel), blk)

    def pack_group(self, grp):
        if grp in self._pack_tabs:
            K.pack_program(*self._pack_tabs[grp])
        if grp in self._split_tabs:
            K.split_bf_program(*self._split_tabs[grp])

    def pack_all(self):
        for grp in sorted(set(self._pack_tabs) | set(self._split_tabs)):
            self.pack_group(grp)

    # Training-mode plans re-pack their operands from the parameters at the start of every forward pass (Adam has just changed them).
    # An EVAL-mode network -- the frozen teacher recogniser of the TPGSR step, everything under TextSREvaluator -- keeps its
    # parameters between calls, so its pack + split (94 + 36 us per step for the teacher) run only when the parameters changed:
    # torch bumps a parameter's version counter on every in-place write through the tensor (load_state_dict, copy_, mul_), and whoever
    # writes the arena through raw pointers or through the flat buffer (FusedAdam, a flat broadcast) bumps `_kernel_writes`.
    def _param_key(self):
        # (the parameters are `.data` views of the arena: each carries its OWN version counter, the arena's does not see their writes)
        # (+ whether the bf16 twins exist: an engine bound under "f32" grows them when a split-operand policy records its first plan)
        return (sum(p._version for p in self.P.values()), self._kernel_writes, self.arena.flat.data_ptr(), bool(getattr(self, "_split_tabs", None)))

    @staticmethod
    def _plan_splits(plan) -> bool:
        """does this pack plan rebuild the bf16 twins as well (recorded under a split-operand policy) or the fp32 operands only (`f32`)?"""
        v = getattr(plan, "_tpgsr_splits", None)
        if v is None:
            v = any(op[0] == "tpgsr_split_bf_program" for op in plan.ops)
            try:
                plan._tpgsr_splits = v
            except AttributeError:
                pass
        return v

    def pack_if_stale(self, pack_plan):
        capturing = (not K.DRYRUN) and torch.cuda.is_current_stream_capturing()   # a captured graph must contain its own pack
        key = self._param_key()
        check = (not K.DRYRUN) and (not capturing) and os.environ.get("TPGSR_PACK_CHECK") == "1"
        # plans (and pack plans) are cached per arithmetic policy: one eval-mode engine may serve an `f32` evaluator (its pack plan has no
        # split program) AND a split-operand train step.  The twins are current only if the LAST pack of these parameters split them too
        # (an f32 pack after load_state_dict stored the key, the next x2 forward skipped its pack and multiplied stale twins)
        splits = self._plan_splits(pack_plan)
        if capturing or key != self._packed_key or (splits and not self._packed_split) or os.environ.get("TPGSR_PACK_ALWAYS") == "1":
            pack_plan.run()
            self._packed_key = None if capturing else key
            self._packed_split = splits
            if check:
                self._packed_sum = self._arena_checksum()
        elif check and getattr(self, "_packed_sum", None) is not None and self._arena_checksum() != self._packed_sum:
            # debugging aid: the pack was skipped on an unchanged key, yet the arena's bits are not those that were packed
            raise RuntimeError(f"{type(self.module).__name__}: the parameters changed behind torch's version counters since the operands were "
                               "packed (a write through p.data / a raw pointer / a flat arena view?) -- call module._engine().invalidate_packed() "
                               "after such writes (TPGSR_PACK_CHECK=1 found this)")

    def _arena_checksum(self):
        """(TPGSR_PACK_CHECK=1 only: one device reduction + a host sync per eval forward) an order-independent fingerprint of the parameter
        arena's BITS: the wrapping int64 sum of its words and of their squares' low bits"""
        w = self.arena.flat.view(torch.int32).to(torch.int64)
        return (int(w.sum().item()), int((w * w).sum().item()))

    def invalidate_packed(self):
        """The parameters changed in a way no version counter sees -- a write through `p.data` (`p.data.copy_/mul_/clamp_`: EMA, weight
        clipping), through a raw pointer or a flat view of the arena, a hipGraph replay containing the optimiser: the next eval-mode
        forward re-packs its operands.  FusedAdam, the flat broadcasts and the train steps' `replay()` call this themselves
        (`load_state_dict` / `p.copy_` bump torch's version counters, which the key sums); code that writes parameters behind torch's back must too (`module._engine().invalidate_packed()`).
        TPGSR_PACK_ALWAYS=1 re-packs on every forward; TPGSR_PACK_CHECK=1 fingerprints the arena at every pack and raises when a skipped pack
        finds other bits (debugging aids)."""
        self._kernel_writes += 1
        self._packed_key = None

    def note_packed(self):
        """a training-mode forward has just packed the current parameters (under the policy it was recorded with)"""
        self._packed_key = self._param_key()
        self._packed_split = bool(K.CONV_TERMS)

    def bind(self, device):
        rebuilt = self.arena.ensure(device)
        if not rebuilt and self.device == device:
            return
        self.device = device
        self._plans.clear()
        self._packed_key = None          # the packed operands are rebuilt below: nothing is packed yet
        self._scratch.clear()
        self._live.clear()
        self._pack = []
        self._operands = []
        self._bn_layers = []
        m = self.module
        self.P = dict(m.named_parameters())
        self.B = dict(m.named_buffers())
        self.G = {}
        for name, p in self.P.items():
            o = self.arena.offsets[name]
            self.G[name] = self.arena.grad[o:o + p.numel()]
        for name, b in self.B.items():
            if b.device != device:
                raise RuntimeError(f"buffer {name} is on {b.device}, parameters on {device}: call module.to(device) first")
        self._build_layers()
        self._finish_pack_table()
        self._finish_split_table()

    def flush_counters(self):
        if self._pending_batches and self.device is not None:
            for n, b in self.B.items():
                if n.endswith("num_batches_tracked"):
                    b += self._pending_batches
        self._pending_batches = 0

    def _two_pass(self, key, record):
        """pass 1 sizes the stream-ordered scratch buffers, pass 2 records against their final addresses"""
        if key not in self._plans:
            if K.POLICY != "f32" and self._operands and not self._split_tabs:
                self._finish_split_table()      # bound under "f32" (no bf16 twins), now recording a split-operand policy
            ws = _Ws(self.device)
            record(ws, False)
            self._plans[key] = record(ws, True)
        return self._plans[key]


class TSRNEngine(_EngineBase):
    """Owns the arenas, packed operands, workspaces and the recorded plans of one TSRN / TSRN_TL module."""

    STN_POOLS = [(2, 2), (2, 2), (2, 2), (2, 2), (1, 2), (1, 1)]

    def __init__(self, module: torch.nn.Module, grid_align_corners: bool = False):
        super().__init__(module)
        self.grid_align_corners = grid_align_corners
        # weight-gradient launches on a second stream (Plan.side).  Needs every buffer a wgrad reads to be written once
        # per backward pass: _record_bwd gives each layer its own dy / du / dgi / dgh instead of recycling one set.
        self.overlap_wgrad = os.environ.get("TPGSR_OVERLAP_WGRAD", "1") != "0"
        # all weight-gradient slab reduces of a backward pass in one launch (each layer then keeps its own slab buffers)
        self.defer_reduce = os.environ.get("TPGSR_DEFER_REDUCE", "1") != "0"
        # the STN head's backward (a chain of ~60 small launches that only produces parameter gradients) on a third stream, next to
        # the rest of the step (InfoGen backward -> text-prior generator backward on the caller's stream)
        self.leaf_stn = os.environ.get("TPGSR_LEAF_STN", "1") != "0"
        # ... and with it everything after the last use of the caller's stream's data in the trunk: block 0's convolution gradients,
        # block1 (TPGSR_LEAF_EARLY=0: only the STN head)
        # (measured on MI355X: 7.60 vs 7.55 ms per C3 step -- the longer leaf chain, whose weight gradients run in order with it, lands
        #  on the step's tail -- so it is opt-in: TPGSR_LEAF_EARLY=1)
        self.leaf_early = os.environ.get("TPGSR_LEAF_EARLY", "0") == "1"
        # the text strip's data gradient (a 192 -> 32 projection + the sum over H, per residual block) on the leaf stream (TPGSR_LEAF_STRIP=0: caller's stream)
        self.leaf_strip = os.environ.get("TPGSR_LEAF_STRIP", "1") != "0"
        # InfoGen's forward pass on the weight-gradient stream next to block 0's convolutions: opt-in (TPGSR_SIDE_INFOGEN=1) -- measured
        # 5.614 / 5.598 vs 5.614 / 5.618 ms per C3 step (gpurun_out/r05w): its tile-loop workgroups sit on CUs the trunk's whole-CU
        # convolutions then wait for, and what the caller's stream saves it loses again
        self.side_infogen = os.environ.get("TPGSR_SIDE_INFOGEN", "0") == "1"

    def _build_layers(self):
        m = self.module
        self.in_planes = m.in_planes
        self.srb = m.srb_nums
        self.stn = bool(m.stn)
        self.C = self.P["block1.0.weight"].shape[0]
        self.block1 = ConvLayer(self, "block1.0.weight", "block1.0.bias", 9, 9, 4, 4, need_dgrad=False)
        self.block1_dx = FoldedDgrad(self, "block1.0.weight") if self.stn else None      # its data gradient: only the STN head asks for it
        self.tl = hasattr(m, "infoGen")
        self.rrb = []
        for i in range(self.srb):
            p = f"block{i + 2}"
            g1 = GruLayer(self, p + ".gru1", axis=1)
            self.rrb.append(dict(
                conv1=ConvLayer(self, p + ".conv1.weight", p + ".conv1.bias", 3, 3, 1, 1), bn1=BNLayer(self, p + ".bn1"),
                conv2=ConvLayer(self, p + ".conv2.weight", p + ".conv2.bias", 3, 3, 1, 1),
                bn2=BNLayer(self, p + ".bn2", pad_to=g1.Cin),
                gru1=g1, gru2=GruLayer(self, p + ".gru2", axis=0)))
        if self.tl:
            cfg = [(2, 1), (2, 1), (2, 1), (1, 0)]       # (stride_w, pad_w) of tconv1..4 (model/tsrn.py:89-98)
            self.ig = [TConvStrip(self, f"infoGen.tconv{i + 1}.weight", sw, pw) for i, (sw, pw) in enumerate(cfg)]
            self.ig_bn = [BNLayer(self, f"infoGen.bn{i + 1}") for i in range(4)]
            self.Ct = self.ig[3].Cout
            self.emb_cls = self.ig[0].Cin
        k7 = f"block{self.srb + 2}"
        self.conv7 = ConvLayer(self, k7 + ".0.weight", k7 + ".0.bias", 3, 3, 1, 1)
        self.bn7 = BNLayer(self, k7 + ".1")
        k8 = f"block{self.srb + 3}"
        self.n_up = len([k for k in self.P if k.startswith(k8 + ".") and k.endswith(".conv.weight")])
        if self.n_up != 1:
            raise NotImplementedError("scale_factor != 2 is not on the TPGSR hot path (reference configs use 2)")
        self.up = ConvLayer(self, k8 + ".0.conv.weight", k8 + ".0.conv.bias", 3, 3, 1, 1)
        self.tail = ConvLayer(self, f"{k8}.{self.n_up}.weight", None, tail=True)
        self.tail_bias = self.P[f"{k8}.{self.n_up}.bias"]
        if self.stn:
            self.stn_convs, self.stn_bns = [], []
            for i in range(6):
                cp = f"stn_head.stn_convnet.{2 * i}"
                self.stn_convs.append(ConvLayer(self, cp + ".0.weight", cp + ".0.bias", 3, 3, 1, 1, need_dgrad=(i > 0)))
                self.stn_bns.append(BNLayer(self, cp + ".1"))
            # fc1 sees the NCHW flatten (c*2 + w) of a [N][256][1][2] map == a valid 1x2 conv over NHWC [N][1][2][256]
            w1 = self.P["stn_head.stn_fc1.0.weight"]
            self.fc1 = _FC1AsConv(self, "stn_head.stn_fc1.0.weight", "stn_head.stn_fc1.0.bias", w1.shape[0], w1.shape[1] // 2)
            self.bnf = BNLayer(self, "stn_head.stn_fc1.1")
            self.fc2 = ConvLayer(self, "stn_head.stn_fc2.weight", "stn_head.stn_fc2.bias", wscale=0.1)  # fc2(0.1*feat), stn_head.py:100
            self.tps_inv = self.B["tps.inverse_kernel"]
            self.tps_repr = self.B["tps.target_coordinate_repr"]
            self.NC = self.B["tps.target_control_points"].shape[0]

    # ------------------------------------------------------------------------------------------------------------
    def plans(self, N, H, W, training, slot=0, defer_join=False):
        """slot: independent activation workspace (a shared SR net runs once per cascade stage, each stage's backward
        needs its own saved activations -- interfaces/super_resolution.py:306-385 with --sr_share).
        defer_join: the backward plan does NOT end by joining the weight-gradient stream -- the caller orders its stream after the
        side stream before anything reads the parameter gradients (TPGSRTrainStep at world size 1: the student's backward pass
        starts while this network's weight gradients are still running; ONE join before the optimiser)"""
        return self._two_pass((N, H, W, bool(training), slot, bool(defer_join), K.POLICY),
                              lambda ws, final: self._record(N, H, W, training, ws, final, bool(defer_join)))

    def _record(self, N, H, W, training, ws, final, defer_join=False):
        pre, fwd, bwd, pack = Plan("tsrn_fwd_pre"), Plan("tsrn_fwd"), Plan("tsrn_bwd"), Plan("tsrn_pack")
        pre.final = fwd.final = bwd.final = pack.final = final
        bwd.overlap = self.overlap_wgrad
        fwd.overlap = self.overlap_wgrad      # (its one side section: InfoGen's forward pass)
        bwd.deferred = [] if self.defer_reduce else None
        bwd.use_leaf = self.leaf_stn and self.overlap_wgrad and self.defer_reduce
        self._cur_ws, self._wg_idx, self._compose = ws, 0, []
        for bn in self._bn_layers:
            bn.use(ws)
        # the forward pass in two plans: everything that does not depend on the text prior (operand packing, the STN head + TPS
        # rectification, block1) -- a train step launches it on another stream next to the text-prior generator's forward pass
        # (forward_pre) -- and the rest
        if not training:
            with recording(pack):
                self.pack_all()
        with recording(pre), K.conv_terms(K.terms_for("sr", "fwd")):
            b1 = self._record_fwd_pre(N, H, W, training, ws)
        with recording(fwd), K.conv_terms(K.terms_for("sr", "fwd")):
            self._record_fwd(N, H, W, training, ws, b1)
        if training:
            with recording(bwd), K.conv_terms(K.terms_for("sr", "bwd")):
                self._record_bwd(N, H, W, ws)
                if self.defer_reduce:
                    # the leaf stream's slabs are reduced on the leaf stream (TPGSR_SPLIT_LEAF_REDUCE=0: one program on the
                    # weight-gradient stream, which then waits for the leaf chain)
                    K.flush_wgrad_reduces(split_leaf=os.environ.get("TPGSR_SPLIT_LEAF_REDUCE", "1") != "0")
                else:
                    bwd.leaf_to_side()
                self.flush_compose_bwd()
                if not defer_join:
                    bwd.join()
        return dict(pre=pre, fwd=fwd, bwd=bwd, pack=pack, ws=ws)

    # ---- forward -------------------------------------------------------------------------------------------------
    def _record_fwd_pre(self, N, H, W, training, ws):
        """model/tsrn.py:183-186 (STN, training only) and block1: independent of the text prior"""
        Cc, Ci = self.C, self.in_planes
        P1 = N * H * W
        if training:
            self.pack_all()          # (eval mode: its own plan, run only when the parameters changed -- pack_if_stale)
        x = ws("x_nhwc", P1, Ci)
        K.nchw_to_nhwc(K.DynPtr("x"), N, Ci, H, W, x)
        xin = x
        if self.stn and training:
            xin = self._record_stn_fwd(N, H, W, x, ws)
        c1 = ws("c1", P1, Cc)
        b1 = ws("b1", P1, Cc)
        self.block1.fwd(N, H, W, xin, c1)
        K.prelu_fwd(c1, self.P["block1.1.weight"], P1 * Cc, b1)
        return b1

    def _record_fwd(self, N, H, W, training, ws, b1):
        Cc, Ci = self.C, self.in_planes
        P1 = N * H * W
        temb = None
        if self.tl:
            # InfoGen (four strip convolutions + BatchNorms + the resample: ~160 us of small launches) only depends on the text prior, and
            # nothing needs its output before block 0's first GruBlock: it CAN run on the weight-gradient stream -- idle during a forward
            # pass -- next to block 0's two 3x3 convolutions, the caller's stream joining in front of that GruBlock (side_infogen: off by
            # default, it bought nothing)
            with (K.side() if self.side_infogen else contextlib.nullcontext()):
                temb = self._record_infogen_fwd(N, W, training, ws)
        cur = b1
        for i, L in enumerate(self.rrb):
            t = f"r{i}_"
            y1, y2 = ws(t + "y1", P1, Cc), ws(t + "y2", P1, Cc)
            h1, out = ws(t + "h1", P1, Cc), ws(t + "out", P1, Cc)
            gi1, gi2 = (lambda t=t: ws(t + "gi1", P1, 192)), (lambda t=t: ws(t + "gi2", P1, 192))      # only without the one-launch GruBlock
            gt1 = ws(t + "gt1", P1, 256) if training else None      # GRU gate values, kept for back-propagation
            gt2 = ws(t + "gt2", P1, 256) if training else None
            part, _ = L["bn1"].partial(P1)
            fin = L["bn1"].fin(P1, L["conv1"].b) if training else None      # finalized by the convolution's own launch
            g1 = L["conv1"].fwd(N, H, W, cur, y1, bn_partial=part if training else None, bn_fin=fin, bn_coarse=training)
            a1 = ws(t + "a1", P1, Cc)                   # mish(bn1(y1)) once: a 3x3 consumer would re-apply it 9x per element
            dd = L["bn1"].derive_fwd(P1, L["conv1"].b, g1.bn_row_tiles) if (training and fin is None) else None
            if dd is not None:                          # the materialising launch finalizes bn1 itself
                K.affine_act_bnd(dd, y1, P1, "mish", a1)
            else:
                if fin is None:
                    L["bn1"].finalize(P1, L["conv1"].b, training, g1.bn_row_tiles)
                K.affine_act(y1, P1, Cc, L["bn1"].scale, L["bn1"].shift, "mish", a1)
            fin = L["bn2"].fin(P1, L["conv2"].b) if training else None
            g2 = L["conv2"].fwd(N, H, W, a1, y2, bn_partial=part if training else None, bn_fin=fin, bn_coarse=training)
            if fin is None:
                L["bn2"].finalize(P1, L["conv2"].b, training, g2.bn_row_tiles)
            if self.tl:   # torch.cat([bn2(y2), text strip], 1) inside the 1x1 conv's loader (model/tsrn.py:419-423)
                if i == 0 and self.side_infogen and K._REC is not None and K._REC.forks:
                    K._REC.join()          # the text strip comes off the other stream
                L["gru1"].fwd(N, H, W, y2, gi1, h1, gt1, in_b=temb, cin_a=Cc, **L["bn2"].loader)
            else:
                L["gru1"].fwd(N, H, W, y2, gi1, h1, gt1, **L["bn2"].loader)
            L["gru2"].fwd(N, H, W, cur, gi2, out, gt2, in2=h1)
            cur = out
        if K._REC is not None and K._REC.forks:
            K._REC.join()                  # (a network without residual blocks: nobody has waited for InfoGen yet)
        y7 = ws("y7", P1, Cc)
        part, _ = self.bn7.partial(P1)
        fin = self.bn7.fin(P1, self.conv7.b) if training else None
        g7 = self.conv7.fwd(N, H, W, cur, y7, bn_partial=part if training else None, bn_fin=fin, bn_coarse=training)
        if fin is None:
            self.bn7.finalize(P1, self.conv7.b, training, g7.bn_row_tiles)
        ups = ws("ups", 4 * P1, Cc)                      # pre-mish, pixel-shuffled [N][2H][2W][C]
        self.up.fwd(N, H, W, y7, ups, in2=b1, out_ps=True, **self.bn7.loader)
        mu = ws("mups", 4 * P1, Cc)                      # mish(ups) once (the 9-tap tail conv and its wgrad both read it)
        K.affine_act(ups, 4 * P1, Cc, None, None, "mish", mu)
        Pt = ws("Pt", 4 * P1, self.tail.Cout)
        self.tail.fwd(N, 2 * H, 2 * W, mu, Pt)
        K.tail_shiftsum_tanh(Pt, self.tail_bias, N, 2 * H, 2 * W, self.tail.Co, self.tail.KS, K.DynPtr("sr"))

    def _ig_widths(self, Wp):
        ws_ = [Wp]
        for tc in self.ig:
            ws_.append(tc.out_w(ws_[-1]))
        return ws_

    def _record_infogen_fwd(self, N, W, training, ws, Wp=26):
        """InfoGen (4 x ConvTranspose2d+BN+ReLU on the H=1 strip) + F.interpolate(..., bilinear, align_corners=True)
        -> text strip [N][W][Ct] (all H rows of the reference's spatial_t_emb are identical)."""
        pri = ws("prior_nhwc", N * Wp, self.emb_cls)
        K.nchw_to_nhwc(K.DynPtr("prior"), N, self.emb_cls, 1, Wp, pri)
        if self.ig[0].Cp != self.emb_cls:     # 37 classes -> 40 channels: tconv1 stays on the 16-byte loaders
            prp = ws("prior_p", N * Wp, self.ig[0].Cp)
            K.pad_channels(pri, N * Wp, self.emb_cls, self.ig[0].Cp, prp)
            pri = prp
        widths = self._ig_widths(Wp)
        cur, loader = pri, {}
        for i, (tc, bn) in enumerate(zip(self.ig, self.ig_bn)):
            out = ws(f"ig_t{i}", N * widths[i + 1], tc.Cout)
            part, _ = bn.partial(N * widths[i + 1])
            tc.fwd(N, widths[i], cur, out, bn_partial=part if training else None, **loader)
            bn.finalize(N * widths[i + 1], None, training)
            cur, loader = out, dict(in_act="relu", **bn.loader)
        temb = ws("temb", N * W, self.Ct)
        K.strip_resample_fwd(cur, self.ig_bn[3].scale, self.ig_bn[3].shift, "relu", N, widths[4], W, self.Ct, temb)
        return temb

    def _record_infogen_bwd(self, N, W, ws, Wp=26):
        t = ws.t
        widths = self._ig_widths(Wp)
        dz = ws("ig_dz3", N * widths[4], self.Ct)
        K.strip_resample_bwd(t["ig_t3"], self.ig_bn[3].scale, self.ig_bn[3].shift, "relu", t["dtemb"], N, widths[4], W, self.Ct, dz)
        da, act, fz = dz, "none", None
        for i in range(3, -1, -1):
            tc, bn = self.ig[i], self.ig_bn[i]
            M = N * widths[i + 1]
            dy = ws(f"ig_dy{i}", M, tc.Cout)
            bn.backward(da, None, t[f"ig_t{i}"], M, act, dy, fused=fz)
            xin = t[f"ig_t{i - 1}"] if i > 0 else t.get("prior_p", t["prior_nhwc"])
            loader = dict(in_act="relu", **self.ig_bn[i - 1].loader) if i > 0 else {}
            tc.wgrad(N, widths[i], xin, dy, loader=loader)
            da = ws(f"ig_da{i}", N * widths[i], tc.Cin)
            act = "relu"
            fz = self.ig_bn[i - 1].fuse_stats(t[f"ig_t{i - 1}"], N * widths[i], act, tc.Cout) if i > 0 else None
            tc.dgrad(N, widths[i], dy, da, bnb=fz)
        K.nhwc_to_nchw(da, N, self.emb_cls, 1, Wp, K.DynPtr("dprior"))

    def _stn_dims(self, H, W):
        dims = []
        h, w = H, W
        for ph, pw in self.STN_POOLS:
            dims.append((h, w))
            h, w = h // ph, w // pw
        if (h, w) != (1, 2):
            raise ValueError(f"STN head needs a {16}x{64} low-resolution input (got {H}x{W}): its fc1 expects a 1x2 map")
        return dims

    def _record_stn_fwd(self, N, H, W, x, ws):
        dims = self._stn_dims(H, W)
        cur = x
        for i, (conv, bn, (h, w), (ph, pw)) in enumerate(zip(self.stn_convs, self.stn_bns, dims, self.STN_POOLS)):
            M = N * h * w
            s = ws(f"stn_s{i}", M, conv.Cout)
            part, _ = bn.partial(M)
            conv.fwd(N, h, w, cur, s, bn_partial=part)
            dd = bn.derive_fwd(M, conv.b) if i < 5 else None
            if dd is None:
                bn.finalize(M, conv.b, True)
            if i < 5:
                a = ws(f"stn_a{i}", N * (h // ph) * (w // pw), conv.Cout)
                if dd is not None:                      # the pooling launch finalizes this stage's BatchNorm itself
                    K.affine_act_pool_bnd(dd, s, N, h, w, "relu", ph, pw, a)
                else:
                    K.affine_act_pool(s, N, h, w, conv.Cout, bn.scale, bn.shift, "relu", ph, pw, a)
                cur = a
            else:
                cur = s  # relu(bn(.)) of the last stage rides on fc1's loader
        f1 = ws("stn_f1", N, self.fc1.Cout)
        part, _ = self.bnf.partial(N)
        self.fc1.fwd(N, 1, 2, cur, f1, in_act="relu", bn_partial=part, **self.stn_bns[5].loader)
        self.bnf.finalize(N, self.fc1.b, True)
        ctrl = ws("stn_ctrl", N, 2 * self.NC)
        self.fc2.fwd(N, 1, 1, f1, ctrl, in_act="relu", **self.bnf.loader)
        grid, src = ws("stn_grid", N, H * W, 2), ws("stn_src", N, H * W, 2)
        K.tps_grid_fwd(ctrl, self.tps_inv, self.tps_repr, N, H * W, self.NC, grid, src)
        xr = ws("xr", N * H * W, self.in_planes)
        K.grid_sample_fwd(x, grid, N, H, W, self.in_planes, H, W, self.grid_align_corners, xr)
        return xr

    # ---- backward ------------------------------------------------------------------------------------------------
    def _record_bwd(self, N, H, W, ws):
        Cc, Ci = self.C, self.in_planes
        P1, P4 = N * H * W, 4 * N * H * W
        H2, W2 = 2 * H, 2 * W
        t = ws.t
        tl = self.tail
        # tail: dP, bias grad, weight grad, d mish(ups)
        sb = K.side_batch_begin()      # the tail's / upsample block's / block 7's weight gradients behind one fork
        nblk = K.tail_bwd_blocks(N, H2, W2, tl.Co, tl.KS)
        dPt = ws("dPt", P4, tl.Cout)
        dbp = self.scratch("tail_dbp", nblk * tl.Co)
        K.tail_bwd(K.DynPtr("sr"), K.DynPtr("dsr"), N, H2, W2, tl.Co, tl.KS, dPt, dbp, nblk)
        with K.side():   # bias gradient: a leaf, off the critical path
            K.reduce_partials(dbp, nblk, tl.Co, self.G[tl.wname.replace(".weight", ".bias")], accumulate=True)
        tl.wgrad(N, H2, W2, t["mups"], dPt)
        dm = ws("d_ups", P4, Cc)
        if K.bnb_fusable(tl.Cout):     # d(ups) = mish'(ups) * (data gradient), the factor applied in the convolution's epilogue
            tl.dgrad(N, H2, W2, dPt, dm, bnb=dict(y=t["ups"], act="mish", store_dz=True))
        else:
            tl.dgrad(N, H2, W2, dPt, dm)
            K.act_bwd(t["ups"], dm, P4 * Cc, "mish", dm)                  # in place: d(ups), pixel-shuffled layout
        # upsample conv: input was bn7(y7) + b1
        self.up.wgrad(N, H, W, t["y7"], dm, loader=dict(in2=t["b1"], **self.bn7.loader), dy_kw=dict(dy_ps=True))
        d_s = ws("d_s", P1, Cc)                                           # = d(bn7 out) = one of b1's gradients
        fz7 = self.bn7.fuse_stats(t["y7"], P1, "none", self.up.Cout)
        self.up.dgrad(N, H, W, dm, d_s, in_ps=True, bnb=fz7)
        uniq = self.overlap_wgrad

        def buf(name, tag, C_):    # one buffer per use when a side-stream wgrad reads it, else one recycled buffer
            return ws(tag + name if uniq else name, P1, C_)

        dy = buf("dy", "b7_", Cc)
        gA, gB = ws("gA", P1, Cc), ws("gB", P1, Cc)
        last_out = t[f"r{self.srb - 1}_out"] if self.srb else t["b1"]
        if fz7 is not None and self.bn7.can_fold(P1) and self.conv7.dgrad_takes_folded_apply(N, H, W, d_s, t["y7"], self.bn7.coef):
            # the apply (dy for the weight gradient) on the weight-gradient stream; the data gradient reads d_s through the folded loader
            ld7 = self.bn7.backward_folded(d_s, t["y7"], P1, dy, fz7)
            self.conv7.wgrad(N, H, W, last_out, dy)
            self.conv7.dgrad(N, H, W, d_s, gA, **ld7)
        else:
            self.bn7.backward(d_s, None, t["y7"], P1, "none", dy, fused=fz7)
            self.conv7.wgrad(N, H, W, last_out, dy)
            self.conv7.dgrad(N, H, W, dy, gA)
        have_B = False
        da = ws("da", P1, Cc)
        leaf = contextlib.ExitStack()
        K.side_batch_end(sb)
        nbb = max(1, int(os.environ.get("TPGSR_SIDE_BATCH_BLOCKS", "2")))     # blocks per side batch (measured: 1 -> 7.11, 2 -> 7.085, 5 -> 7.40 ms)
        sb = False
        for i in range(self.srb - 1, -1, -1):
            L = self.rrb[i]
            if (self.srb - 1 - i) % nbb == 0 and not (i == 0 and self.leaf_early):
                sb = K.side_batch_begin()   # one fork per block (or per nbb blocks) instead of four
            p = f"r{i}_"
            dgi, dgh = buf("dgi", p + "g2_", 192), buf("dgh", p + "g2_", 192)
            X = t[f"r{i - 1}_out"] if i > 0 else t["b1"]
            y1, y2, gt1, h1, gt2, out = (t[p + n] for n in ("y1", "y2", "gt1", "h1", "gt2", "out"))
            # gru2 (input X + h1): parameter grads + d(X + h1) -> gA (incoming gA/gB are dead after the scan)
            L["gru2"].bwd(N, H, W, X, gt2, out, gA, gB if have_B else None, dgi, dgh, gA, in2=h1)
            if sb and os.environ.get("TPGSR_SIDE_BATCH_SPLIT", "0") == "1":
                # experiment: what is held so far (+ gru2's weight gradients) goes out here, next to gru1's BiGRU backward kernel
                K.side_batch_end(sb)
                sb = K.side_batch_begin()
            # gru1 (input bn2(y2) [+ text strip]): dh = gA
            dgi, dgh = buf("dgi", p + "g1_", 192), buf("dgh", p + "g1_", 192)
            if self.tl:
                g1 = L["gru1"]
                g1.bwd(N, H, W, y2, gt1, h1, gA, None, dgi, dgh, None, in_b=t["temb"], cin_a=Cc, **L["bn2"].loader)
                # data gradient of the composed 96->192 projection in two column blocks: image features and text strip
                # (leaf_early: block 0's BatchNorm backward runs on the leaf stream, its producer here -- no shared scratch across streams)
                fz2 = None if (i == 0 and self.leaf_early) else L["bn2"].fuse_stats(y2, P1, "none", 192)
                fold2 = fz2 is not None and L["bn2"].can_fold(P1) and L["conv2"].dgrad_takes_folded_apply(N, H, W, da, y2, L["bn2"].coef)
                dz2 = ws(p + "c2_dz", P1, Cc) if fold2 else da      # (folded: the weight-gradient stream reads it later -- a buffer of its own)
                K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, 192, Cc), dgi, g1.wc_d, dz2, wt_ld=g1.Cin, wt_coff=0, bnb=fz2))
                # the text strip's share of the gradient (summed over the H rows it was broadcast to, accumulated over the blocks) only meets
                # the caller's stream again at the InfoGen backward pass: it runs on the leaf stream (idle until the STN head's backward),
                # block after block in order, next to the rest of this block on the caller's stream (round 5: -18 us per block there)
                with (K.leaf() if self.leaf_strip else contextlib.nullcontext()):
                    dtb = ws("d_tb", P1, self.Ct)
                    K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, 192, self.Ct), dgi, g1.wc_d, dtb, wt_ld=g1.Cin, wt_coff=Cc))
                    K.hsum(dtb, N, H, W, self.Ct, ws("dtemb", N * W, self.Ct), accumulate=(i != self.srb - 1))
            else:
                fz2 = None if (i == 0 and self.leaf_early) else L["bn2"].fuse_stats(y2, P1, "none", 192)
                fold2 = fz2 is not None and L["bn2"].can_fold(P1) and L["conv2"].dgrad_takes_folded_apply(N, H, W, da, y2, L["bn2"].coef)
                dz2 = ws(p + "c2_dz", P1, Cc) if fold2 else da
                L["gru1"].bwd(N, H, W, y2, gt1, h1, gA, None, dgi, dgh, dz2, dx_bnb=fz2, **L["bn2"].loader)
            if i == 0 and self.leaf_early:
                # Everything below only feeds parameter gradients (block 0's convolutions, block1, the STN head): the text-strip gradient
                # dtemb is final here, so the caller's stream goes straight on to the InfoGen backward and the text-prior generator's
                # backward pass while this tail runs on the leaf stream.  With the text strip's gradient on the leaf stream (leaf_strip) the
                # caller's stream must first see block 0's strip section finish: the 2 -> 0 edge goes in HERE, while this recording is
                # still on the caller's stream (leaf_join() is a no-op once the leaf section is open, and the InfoGen backward reads dtemb
                # on the caller's stream after the section closes -- ADVICE round 5)
                if self.tl and self.leaf_strip:
                    K.leaf_join()
                leaf.enter_context(K.leaf())
            dy = buf("dy", p + "c2_", Cc)
            fold1 = (K.bnb_fusable(L["conv2"].Cout) and not (i == 0 and self.leaf_early) and L["bn1"].can_fold(P1) and
                     L["conv1"].dgrad_takes_folded_apply(N, H, W, da, y1, L["bn1"].coef))
            fz1 = L["bn1"].fuse_stats(y1, P1, "mish", L["conv2"].Cout, store_dz=fold1)
            dz1 = ws(p + "c1_dz", P1, Cc) if fold1 else da
            if fold2:
                # bn2's apply runs on the weight-gradient stream (dy for conv2's weight gradient); conv2's data gradient reads dz2 + y2 with
                # the three coefficients in its loader (csrc/conv_loader.h, LD bit 32): one launch less on the caller's stream per BatchNorm
                ld2 = L["bn2"].backward_folded(dz2, y2, P1, dy, fz2)
                L["conv2"].wgrad(N, H, W, t[p + "a1"], dy)
                L["conv2"].dgrad(N, H, W, dz2, dz1, bnb=fz1, **ld2)      # d mish(bn1(y1)) (fold1: times mish', i.e. dz of bn1)
            else:
                L["bn2"].backward(dz2, None, y2, P1, "none", dy, fused=fz2)
                L["conv2"].wgrad(N, H, W, t[p + "a1"], dy)
                L["conv2"].dgrad(N, H, W, dy, dz1, bnb=fz1)                # d mish(bn1(y1))
            dy = buf("dy", p + "c1_", Cc)
            if fold1:
                ld1 = L["bn1"].backward_folded(dz1, y1, P1, dy, fz1)
                L["conv1"].wgrad(N, H, W, X, dy)
                L["conv1"].dgrad(N, H, W, dz1, gB, **ld1)                  # second gradient path into X
            else:
                L["bn1"].backward(dz1, None, y1, P1, "mish", dy, fused=fz1)
                L["conv1"].wgrad(N, H, W, X, dy)
                L["conv1"].dgrad(N, H, W, dy, gB)                          # second gradient path into X
            have_B = True
            if (self.srb - 1 - i) % nbb == nbb - 1 or i == 0 or (i == 1 and self.leaf_early):
                K.side_batch_end(sb)
                sb = False
        if self.tl and self.leaf_strip:
            # dtemb is complete once the leaf stream is through block 0's strip gradient (recorded ~100 us of caller's-stream work ago).
            # The edge goes HERE, before the STN head's backward chain is queued on the leaf stream: the InfoGen backward pass below must
            # not wait for that
            K.leaf_join()
        early = bool(self.srb and self.leaf_early)      # the leaf section is already open
        # block1's and InfoGen's weight gradients (+ the PReLU slope's reduce) behind one fork at the end of the plan
        sb = K.side_batch_begin() if (os.environ.get("TPGSR_SIDE_BATCH_TAIL", "1") != "0" and not early) else False
        # b1 receives d_s (long skip) + gA (+ gB)
        if have_B:
            K.add(gA, gB, P1 * Cc, gA)
        dc1 = ws("dc1", P1, Cc)
        nb = 1024
        dap = self.scratch("prelu_dap" + K.stream_tag(), nb)
        K.prelu_bwd(t["c1"], self.P["block1.1.weight"], gA, d_s, P1 * Cc, dc1, dap, nb)
        with K.side():
            K.reduce_partials(dap, nb, 1, self.G["block1.1.weight"], accumulate=True)
        xin = t["xr"] if self.stn else t["x_nhwc"]
        self.block1.wgrad(N, H, W, xin, dc1)
        if self.stn:
            if not early:
                leaf.enter_context(K.leaf())
            self._record_stn_bwd(N, H, W, dc1, ws)
        leaf.close()
        if self.tl:
            self._record_infogen_bwd(N, W, ws)
        K.side_batch_end(sb)

    def _record_stn_bwd(self, N, H, W, dc1, ws):
        t = ws.t
        Ci = self.in_planes
        dxr = ws("dxr", N * H * W, Ci)
        self.block1_dx.run(N, H, W, dc1, ws("stn_Pd", N * H * W, self.block1_dx.NP), dxr)
        dgrid = ws("stn_dgrid", N, H * W, 2)
        K.grid_sample_bwd(t["x_nhwc"], t["stn_grid"], dxr, N, H, W, Ci, H, W, self.grid_align_corners, None, dgrid)
        dctrl = ws("stn_dctrl", N, 2 * self.NC)
        K.tps_grid_bwd(dgrid, t["stn_src"], self.tps_inv, self.tps_repr, N, H * W, self.NC, dctrl)
        f1 = t["stn_f1"]
        self.fc2.wgrad(N, 1, 1, f1, dctrl, loader=dict(in_act="relu", **self.bnf.loader))
        dfa = ws("stn_dfa", N, self.fc1.Cout)
        self.fc2.dgrad(N, 1, 1, dctrl, dfa)                                # d relu(bn(f1))  (0.1 already in wt_d)
        df1 = ws("stn_df1", N, self.fc1.Cout)
        self.bnf.backward(dfa, None, f1, N, "relu", df1)
        s5 = t["stn_s5"]
        self.fc1.wgrad(N, 1, 2, s5, df1, loader=dict(in_act="relu", **self.stn_bns[5].loader))
        dims = self._stn_dims(H, W)
        dact = ws("stn_da5", N * 2, self.stn_convs[5].Cout)                # d relu(bn5(s5)), [N][1][2][256]
        self.fc1.dgrad(N, 1, 2, df1, dact)
        for i in range(5, -1, -1):
            conv, bn = self.stn_convs[i], self.stn_bns[i]
            h, w = dims[i]
            ph, pw = self.STN_POOLS[i]
            M = N * h * w
            s = t[f"stn_s{i}"]
            ds = ws(f"stn_ds{i}", M, conv.Cout)
            if i == 5:
                bn.backward(dact, None, s, M, "relu", ds)
            else:
                dz = ws(f"stn_dz{i}", M, conv.Cout)
                K.affine_act_pool_bwd(s, dact, N, h, w, conv.Cout, bn.scale, bn.shift, "relu", ph, pw, dz)
                bn.backward(dz, None, s, M, "none", ds)
            xin = t[f"stn_a{i - 1}"] if i > 0 else t["x_nhwc"]
            conv.wgrad(N, h, w, xin, ds)
            if i > 0:
                dact = ws(f"stn_da{i - 1}", N * h * w, self.stn_convs[i - 1].Cout)
                conv.dgrad(N, h, w, ds, dact)

    # ---- execution -----------------------------------------------------------------------------------------------
    def forward_pre(self, x: torch.Tensor, training: bool, slot: int = 0, defer_join: bool = False):
        """the prior-independent part of the forward pass (operand packing, STN head + rectification, block1) on the CURRENT stream;
        the caller then runs forward(..., pre_done=True) on a stream ordered after it.  `x` must stay alive until then."""
        if x.dim() != 4 or x.shape[1] != self.module.in_planes:
            raise ValueError(f"expected (N, {self.module.in_planes}, H, W) input, got {tuple(x.shape)}")
        if not x.is_cuda and not K.DRYRUN:
            raise RuntimeError("tpgsr_amd runs on the GPU only (no CPU fallback): move the module and inputs to cuda")
        self.bind(x.device)
        N, _, H, W = x.shape
        pl = self.plans(N, H, W, training, slot, defer_join and training)
        if x.dtype != F32 or not x.is_contiguous():
            raise ValueError("forward_pre needs a contiguous fp32 input (it is read asynchronously)")
        pl["pre"].set_ptr("x", x.data_ptr())
        if not training:
            self.pack_if_stale(pl["pack"])
        pl["pre"].run()
        if training:
            self.note_packed()

    def forward(self, x: torch.Tensor, training: bool, prior: Optional[torch.Tensor] = None, slot: int = 0,
                defer_join: bool = False, pre_done: bool = False) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.module.in_planes:
            raise ValueError(f"expected (N, {self.module.in_planes}, H, W) input, got {tuple(x.shape)}")
        if not x.is_cuda and not K.DRYRUN:
            raise RuntimeError("tpgsr_amd runs on the GPU only (no CPU fallback): move the module and inputs to cuda")
        self.bind(x.device)
        N, _, H, W = x.shape
        pl = self.plans(N, H, W, training, slot, defer_join and training)
        x = x.contiguous().float()
        sr = torch.empty(N, self.in_planes, 2 * H, 2 * W, dtype=F32, device=x.device)
        if not pre_done:
            pl["pre"].set_ptr("x", x.data_ptr())
            if not training:
                self.pack_if_stale(pl["pack"])
            pl["pre"].run()
            if training:
                self.note_packed()
        fwd = pl["fwd"]
        fwd.set_ptr("sr", sr.data_ptr())
        if self.tl:
            if prior is None or tuple(prior.shape) != (N, self.emb_cls, 1, 26) or not (prior.is_cuda or K.DRYRUN):
                raise ValueError(f"TSRN_TL needs a CUDA text prior of shape ({N}, {self.emb_cls}, 1, 26)")
            prior = prior.contiguous().float()
            fwd.set_ptr("prior", prior.data_ptr())
        elif prior is not None:
            raise ValueError("this network takes no text prior")
        fwd.run()
        if training:
            self._pending_batches += 1   # num_batches_tracked is bookkeeping only (momentum is fixed): flushed lazily
        return sr


    def backward(self, x_shape, sr: torch.Tensor, dsr: torch.Tensor, slot: int = 0, defer_join: bool = False):
        N, _, H, W = x_shape
        pl = self.plans(N, H, W, True, slot, defer_join)
        self.arena.attach_grads()
        bwd = pl["bwd"]
        dsr = dsr.contiguous().float()
        bwd.set_ptr("sr", sr.data_ptr())
        bwd.set_ptr("dsr", dsr.data_ptr())
        dprior = None
        if self.tl:
            dprior = torch.empty(N, self.emb_cls, 1, 26, dtype=F32, device=dsr.device)
            bwd.set_ptr("dprior", dprior.data_ptr())
        bwd.run()
        return dprior


class _FC1AsConv(ConvLayer):
    """STN fc1 (model/stn_head.py:48): Linear(2*256, 512) over the NCHW flatten of a [N][256][1][2] map, run as a valid
    1x2 conv over the NHWC map: weight[:, c*2 + w] == conv_weight[:, c, 0, w]."""

    def __init__(self, eng, wname, bname, Cout, Cin):
        self.eng, self.wname, self.bname, self.tail, self.wscale = eng, wname, bname, False, 1.0
        self.w = eng.P[wname]
        self.b = eng.P[bname]
        self.Cout, self.Cin, self.KH, self.KW, self.pad_h, self.pad_w = Cout, Cin, 1, 2, 0, 0
        dev = eng.device
        self.wt_f = torch.empty(2 * Cin, Cout, dtype=F32, device=dev)
        self.wt_d = torch.empty(2 * Cout, Cin, dtype=F32, device=dev)
        eng.add_pack(self.w, self.wt_f, self.wt_d, Cout=Cout, Cin=Cin, KH=1, KW=2, kind=0, f_ld=Cout)
        eng.register_operand(self.wt_f, self.wt_d)
