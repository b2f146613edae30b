"""Static execution plans for the CRNN text-prior generator (reference: model/crnn/crnn.py:29-90): seven convs
(three with train-mode BatchNorm) + four max-pools, then two BidirectionalLSTM(256) + Linear stages.

Same machinery as the TSRN engine: NHWC workspaces, BN/ReLU folded into consumer loaders or into the pooling kernel,
all GEMM-shaped work (convs, LSTM input projections, the per-step recurrent projections, embeddings, every weight
gradient) on the fp32-MFMA implicit-GEMM kernel, gate math in small fused kernels, parameter gradients written into
the flat arena.  The sequence tensor is kept batch-major [N][T][C] (== NHWC with H = 1, W = T), so a time step of the
recurrence is a 1x1 "conv" that reads pixel t of every image (negative pad_w selects the column)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import kernels as K
from .engine import BNLayer, ConvLayer, F32, _EngineBase
from .kernels import ConvGeom, Plan, recording


class Conv0Im2col:
    """cnn.conv0 = nn.Conv2d(1, 64, 3, 1, 1) (crnn.py:45-46).  With one input channel the implicit-GEMM loader would gather
    single floats (the scalar loader: 100 us per launch at bs 48), so the 3x3 neighbourhood is written out once as a
    12-channel map (tpgsr_im2col3x3_c1: 9 taps + 3 zero channels) and the conv runs as a 1x1 conv with Cin = 12 on the
    vector loader; weight gradient = the 1x1 conv's (slab rows 9..11 skipped), d gray = col2im of the 1x1 data gradient."""

    def __init__(self, eng, wname, bname):
        self.eng, self.wname, self.bname = eng, wname, bname
        self.w, self.b = eng.P[wname], eng.P[bname]
        assert tuple(self.w.shape[1:]) == (1, 3, 3)
        self.Cout, self.Cin = self.w.shape[0], 12
        dev = eng.device
        self.wt_f = torch.zeros(12, self.Cout, dtype=F32, device=dev)      # rows 9..11 stay zero
        self.wt_d = torch.zeros(self.Cout, 12, dtype=F32, device=dev)      # columns 9..11 stay zero
        eng.add_pack(self.w, self.wt_f, self.wt_d, Cout=self.Cout, Cin=9, KH=1, KW=1, kind=0, f_ld=self.Cout, d_ld=12)
        eng.register_operand(self.wt_f, self.wt_d)

    def fwd(self, N, H, W, gray, col, out, **kw):
        K.im2col3x3_c1(gray, N, H, W, col)
        K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, 12, self.Cout), col, self.wt_f, out, bias=self.b, **kw))

    def wgrad(self, N, H, W, col, dy):
        eng, g = self.eng, ConvGeom(N, H, W, 12, self.Cout)
        Z = K.wgrad_splits(g.M, g.K, g.Cout)
        part, dbp = eng.wgrad_buffers(Z * g.K * g.Cout, Z * g.Cout)
        with K.side():
            K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(g, col), dy, part, dbp))
            K.wgrad_reduce(part, dbp, Z, g, eng.G[self.wname], eng.G[self.bname], accumulate=True, real=(9, 1, 1, 0))

    def dgrad(self, N, H, W, dy, dcol, dgray):
        K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, self.Cout, 12), dy, self.wt_d, dcol))
        K.col2im3x3_c1(dcol, N, H, W, dgray)


class PaddedLinear:
    """nn.Linear(Cin, Cout) with Cout % 4 != 0 (the 37-class embedding, crnn.py:12): both packed operands are zero-padded
    to Cp = 4*ceil(Cout/4) columns / rows so the forward, data-gradient and weight-gradient GEMMs stay on the 16-byte
    loaders.  The output gradient is handed in padded: dy [M][Cp], columns >= Cout zero."""

    def __init__(self, eng, wname, bname):
        self.eng, self.wname, self.bname = eng, wname, bname
        self.w, self.b = eng.P[wname], eng.P[bname]
        self.Cout, self.Cin = self.w.shape
        self.Cp = (self.Cout + 3) // 4 * 4
        dev = eng.device
        self.wt_f = torch.zeros(self.Cin, self.Cp, dtype=F32, device=dev)
        self.wt_d = torch.zeros(self.Cp, self.Cin, dtype=F32, device=dev)
        eng.add_pack(self.w, self.wt_f, self.wt_d, Cout=self.Cout, Cin=self.Cin, kind=0, f_ld=self.Cp)
        eng.register_operand(self.wt_f, self.wt_d)

    def fwd(self, N, H, W, x, out, **kw):
        K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, self.Cin, self.Cout), x, self.wt_f, out, bias=self.b, wt_ld=self.Cp, **kw))

    def dgrad(self, N, H, W, dy_p, dx):
        K.conv_fwd(K.make_conv_args(ConvGeom(N, H, W, self.Cp, self.Cin), dy_p, self.wt_d, dx))

    def wgrad(self, N, H, W, x, dy_p):
        eng, g = self.eng, ConvGeom(N, H, W, self.Cin, self.Cout)
        Z = K.wgrad_splits(g.M, g.K, g.Cout)
        part, dbp = eng.wgrad_buffers(Z * g.K * g.Cout, Z * g.Cout)
        with K.side():
            K.conv_wgrad(K.make_wgrad_args(K.make_conv_args(g, x), dy_p, part, dbp, dy_ld=self.Cp))
            K.wgrad_reduce(part, dbp, Z, g, eng.G[self.wname], eng.G[self.bname], accumulate=True)


class LstmLayer:
    """BidirectionalLSTM (crnn.py:5-26): nn.LSTM(nIn, Hh, bidirectional) + Linear(2*Hh, nOut)."""

    def __init__(self, eng, prefix: str):
        self.eng, self.prefix = eng, prefix
        P, dev = eng.P, eng.device
        r = prefix + ".rnn."
        self.r = r
        self.Hh = Hh = P[r + "weight_hh_l0"].shape[1]
        self.Cin = Cin = P[r + "weight_ih_l0"].shape[1]
        G4 = 4 * Hh
        self.wih_f = torch.empty(Cin, 2 * G4, dtype=F32, device=dev)      # [K=Cin][2*4Hh]
        self.wih_d = torch.empty(2 * G4, Cin, dtype=F32, device=dev)      # dgrad operand
        self.bih = torch.empty(2 * G4, dtype=F32, device=dev)
        self.bhh = torch.empty(2, G4, dtype=F32, device=dev)
        self.whh_f = torch.empty(2, Hh, G4, dtype=F32, device=dev)         # per direction [K=Hh][4Hh] = W_hh^T
        for d, suf in enumerate(("", "_reverse")):
            eng.add_pack(P[r + "weight_ih_l0" + suf], self.wih_f, self.wih_d[d * G4:(d + 1) * G4], Cout=G4, Cin=Cin, kind=0,
                         f_ld=2 * G4, f_coff=d * G4)
            eng.add_pack(P[r + "bias_ih_l0" + suf], self.bih[d * G4:(d + 1) * G4], None, kind=2)
            eng.add_pack(P[r + "bias_hh_l0" + suf], self.bhh[d], None, kind=2)
            eng.add_pack(P[r + "weight_hh_l0" + suf], self.whh_f[d], None, Cout=G4, Cin=Hh, kind=0, f_ld=G4)
        eng.register_operand(self.wih_f, self.wih_d)
        nout = P[prefix + ".embedding.weight"].shape[0]
        self.emb = (ConvLayer if nout % 4 == 0 else PaddedLinear)(eng, prefix + ".embedding.weight", prefix + ".embedding.bias")

    def fwd(self, N, T, x, G, gh, Cst, out, e, **loader):
        """x [N][T][Cin] -> G (gates) -> out [N][T][2Hh] -> e = Linear(out) [N][T][nOut]"""
        Hh, G4, P = self.Hh, 4 * self.Hh, self.eng.P
        K.conv_fwd(K.make_conv_args(ConvGeom(N, 1, T, self.Cin, 2 * G4), x, self.wih_f, G, bias=self.bih, **loader))
        if K.LSTM_SEQ and Hh == 256 and N <= 64 and K.lstm_seq_coresident(self.eng.device):      # the whole recurrence as one persistent launch
            if not hasattr(self, "_seq"):
                self._seq = {}
            gran = K.LSTM_GRANULE and T <= 31
            # one exchange buffer per stream (the teacher runs on its own) AND batch size: rows >= N of a granule buffer are never
            # rewritten, so a buffer shared across batch sizes could hand a larger batch granules whose (epoch, step) tag happens to
            # match again after the 11-bit epoch has wrapped
            key = (K.current_stream().cuda_stream, gran, N)
            if key not in self._seq:
                self._seq[key] = K.lstm_seq_granule_buffers(self.eng.device) if gran else K.lstm_seq_buffers(self.eng.device)
            hx, sync = self._seq[key]
            (K.lstm_seq_fwdg if gran else K.lstm_seq_fwd)(G, self.whh_f, self.bhh, Cst, out, hx, sync, N, T, Hh)
            self.emb.fwd(N, 1, T, out, e)
            return
        if K.LSTM_STEPX and Hh == 256 and N <= 64:           # one fused launch per time step
            if not hasattr(self, "_stepx"):
                self._stepx = {}
            key = K.current_stream().cuda_stream               # per stream (the teacher runs on its own)
            if key not in self._stepx:
                self._stepx[key] = K.lstm_stepx_buffers(self.eng.device)
            wfr, hx = self._stepx[key]
            K.lstm_wfrag(self.whh_f, wfr, Hh)
            for s in range(T):
                K.lstm_stepx_fwd(G, wfr, self.bhh, Cst, out, hx, N, T, Hh, s)
            self.emb.fwd(N, 1, T, out, e)
            return
        S = Hh // (32 * K.LSTM_FWD_KCHUNKS)                   # K-split of the recurrent projection: 2 * S * 4Hh/64 workgroups
        for s in range(T):
            if s > 0:     # gh = h_prev W_hh^T, both directions in one split-K launch (h_prev: time s-1 / T-s of `out`)
                a = [out.data_ptr() + 4 * ((s - 1 if d == 0 else T - s) * 2 * Hh + d * Hh) for d in range(2)]
                K.lstm_rec_gemm(a[0], a[1], T * 2 * Hh, self.whh_f[0], self.whh_f[1], N, Hh, G4, S, gh)
            K.lstm_step_fwd(G, gh if s > 0 else None, S, self.bhh, Cst, out, N, T, Hh, s)
        self.emb.fwd(N, 1, T, out, e)

    def bwd(self, N, T, x, G, Cst, out, de, dout, dhc, dcc, dx, dx_bnb=None, **loader):
        """de = dL/d e  ->  all parameter gradients, dx = dL/d loader(x) (if dx is not None); dx_bnb: BNLayer.fuse_stats(...) of the
        BatchNorm dx is the incoming gradient of"""
        eng, P, Gd, r = self.eng, self.eng.P, self.eng.G, self.r
        Hh, G4 = self.Hh, 4 * self.Hh
        # (only with deferred slab reduces: every launch then has slab buffers of its own -- the five run concurrently)
        batch = K.LSTM_WGRAD_BATCH and K.CONV_TERMS > 0 and K.deferring() and isinstance(self.emb, (ConvLayer, PaddedLinear))
        if not batch:
            self.emb.wgrad(N, 1, T, out, de)
        self.emb.dgrad(N, 1, T, de, dout)
        S = G4 // (32 * K.LSTM_BWD_KCHUNKS)                 # K-split of the recurrent gradient GEMM (K = 4 Hh)
        seq = K.LSTM_SEQ_BWD and Hh == 256 and N <= 64 and K.lstm_seq_coresident(self.eng.device)
        if seq:                                              # the whole BPTT recurrence as one persistent launch
            if not hasattr(self, "_seqb"):
                self._seqb = {}
            gran = K.LSTM_GRANULE_BWD and T <= 255
            key = (K.current_stream().cuda_stream, gran, N)
            if key not in self._seqb:
                self._seqb[key] = K.lstm_seq_bwd_granule_buffers(self.eng.device) if gran else K.lstm_seq_bwd_buffers(self.eng.device)
            px, sync = self._seqb[key]
            (K.lstm_seq_bwdg if gran else K.lstm_seq_bwd)(G, Cst, dout, P[r + "weight_hh_l0"], P[r + "weight_hh_l0_reverse"], px, sync, N, T, Hh)
        for s in range(0 if seq else T):
            if s > 0:     # dh_prev = dG[t_next] W_hh (operand [K=4Hh][Hh] is the PyTorch weight itself), split over K
                a = [G.data_ptr() + 4 * (((T - s if d == 0 else s - 1) * 2 + d) * G4) for d in range(2)]
                K.lstm_rec_gemm(a[0], a[1], T * 2 * G4, P[r + "weight_hh_l0"], P[r + "weight_hh_l0_reverse"], N, G4, Hh, S, dhc)
            K.lstm_step_bwd(G, Cst, dout, dhc if s > 0 else None, S, dcc, N, T, Hh, s)
        with K.side():     # weight gradients: leaves of the graph, on the side stream once the gate gradients are final
            # the five GEMMs of the layer -- 2 directions x (hidden side, input side) + the embedding, M = N T rows each -- go out as ONE
            # launch (tpgsr_conv_wgrad_batch; the input side of the first layer, behind the CNN's BatchNorm + ReLU loader, as a second):
            # alone each is 25-65 us of start-up, and the ten of the two layers sit at the very end of the training step
            wargs, reduces = [], []
            if batch:
                emb = self.emb
                ge = ConvGeom(N, 1, T, emb.Cin, emb.Cout)
                Z = K.wgrad_splits(ge.M, ge.K, ge.Cout)
                part, dbp = eng.wgrad_buffers(Z * ge.K * ge.Cout, Z * ge.Cout)
                dy_kw = dict(dy_ld=emb.Cp) if isinstance(emb, PaddedLinear) else {}
                wargs.append(K.make_wgrad_args(K.make_conv_args(ge, out), de, part, dbp, **dy_kw))
                reduces.append((part, dbp, Z, ge, Gd[emb.wname], Gd[emb.bname]))
            for d, suf in enumerate(("", "_reverse")):
                sgn = 1 if d == 0 else -1
                # hidden side: dW_hh[d] = dG[:, d]^T h_prev, db_hh[d] = colsum(dG[:, d])
                gh_ = ConvGeom(N, 1, T, Hh, G4, 1, 1, 0, sgn, 1, T)
                Z = K.wgrad_splits(gh_.M, gh_.K, G4)
                part, dbp = eng.wgrad_buffers(Z * Hh * G4, Z * G4)
                wargs.append(K.make_wgrad_args(K.make_conv_args(gh_, out, in_ld=2 * Hh, in_coff=d * Hh), G, part, dbp,
                                               dy_ld=2 * G4, dy_coff=d * G4))
                reduces.append((part, dbp, Z, gh_, Gd[r + "weight_hh_l0" + suf], Gd[r + "bias_hh_l0" + suf]))
                # input side
                gi_ = ConvGeom(N, 1, T, self.Cin, G4)
                Z = K.wgrad_splits(gi_.M, gi_.K, G4)
                part, dbp = eng.wgrad_buffers(Z * self.Cin * G4, Z * G4)
                wargs.append(K.make_wgrad_args(K.make_conv_args(gi_, x, **loader), G, part, dbp, dy_ld=2 * G4, dy_coff=d * G4))
                reduces.append((part, dbp, Z, gi_, Gd[r + "weight_ih_l0" + suf], Gd[r + "bias_ih_l0" + suf]))
            if batch:
                K.conv_wgrad_batch(wargs)
            else:
                for w_ in wargs:
                    K.conv_wgrad(w_)
            for part, dbp, Z, g_, dw_, db_ in reduces:
                K.wgrad_reduce(part, dbp, Z, g_, dw_, db_, accumulate=True)
        if dx is not None:
            K.conv_fwd(K.make_conv_args(ConvGeom(N, 1, T, 2 * G4, self.Cin), G, self.wih_d, dx, bnb=dx_bnb))


class CRNNEngine(_EngineBase):
    IMG_HW = (32, 100)
    # (kernel, stride, padding) of pooling0..3 (crnn.py:56-66); None = no pooling after that conv
    POOLS = {0: ((2, 2), (2, 2), (0, 0)), 1: ((2, 2), (2, 2), (0, 0)), 3: ((2, 2), (2, 1), (0, 1)), 5: ((2, 2), (2, 1), (0, 1))}
    BN_AT = (2, 4, 6)
    EARLY_CONVS = 3          # convolutions of pack group 0 = the part of the forward plan in front of the cut (TPGSR_CRNN_PACK_SPLIT=0: no cut)

    def _build_layers(self):
        self.convs, self.bns = [], {}
        for i in range(7):
            k, pad = (3, 1) if i < 6 else (2, 0)
            # pack group 0: conv0..conv2 (0.37 M parameters), packed at the start of the forward plan; group 1: everything behind them
            # (8 M), packed by a plan of its own -- a train step runs it on another stream while conv0..conv2 execute (forward(late_stream=))
            self._pack_group = 0 if i < self.EARLY_CONVS else 1
            if i == 0:
                self.convs.append(Conv0Im2col(self, "cnn.conv0.weight", "cnn.conv0.bias"))
            else:
                self.convs.append(ConvLayer(self, f"cnn.conv{i}.weight", f"cnn.conv{i}.bias", k, k, pad, pad, need_dgrad=True))
            if i in self.BN_AT:
                self.bns[i] = BNLayer(self, f"cnn.batchnorm{i}")
        self.lstm = [LstmLayer(self, "rnn.0"), LstmLayer(self, "rnn.1")]
        self._pack_group = 0
        self.nclass = self.P["rnn.1.embedding.weight"].shape[0]

    def _dims(self):
        """(H, W) seen by conv i and after its optional pool"""
        h, w = self.IMG_HW
        dims = []
        for i in range(7):
            k, pad = (3, 1) if i < 6 else (2, 0)
            oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
            ph, pw = oh, ow
            if i in self.POOLS:
                (kh, kw), (sh, sw), (pdh, pdw) = self.POOLS[i]
                ph, pw = (oh + 2 * pdh - kh) // sh + 1, (ow + 2 * pdw - kw) // sw + 1
            dims.append(((h, w), (oh, ow), (ph, pw)))
            h, w = ph, pw
        assert h == 1, "the height of conv must be 1"   # crnn.py:83
        return dims

    def plans(self, N, training, slot=0):
        return self._two_pass((N, bool(training), slot, getattr(self, "role", "tpg"), K.POLICY), lambda ws, final: self._record(N, training, ws, final))

    def _record(self, N, training, ws, final):
        fwd, bwd, bwd_b, dgp, pack = Plan("crnn_fwd"), Plan("crnn_bwd"), Plan("crnn_bwd_b"), Plan("crnn_dgray"), Plan("crnn_pack")
        fwd_b, pack_late = Plan("crnn_fwd_b"), Plan("crnn_pack_late")
        fwd.final = bwd.final = bwd_b.final = dgp.final = pack.final = fwd_b.final = pack_late.final = final
        split_pack = training and os.environ.get("TPGSR_CRNN_PACK_SPLIT", "1") != "0"
        # weight gradients on the side stream + one batched slab reduce, as in TSRNEngine (every buffer a weight-gradient
        # launch reads -- ds{i}, saved activations, the LSTM gate gradients after the time loop -- is written once per pass)
        bwd.overlap = bwd_b.overlap = os.environ.get("TPGSR_OVERLAP_WGRAD", "1") != "0"
        defer = os.environ.get("TPGSR_DEFER_REDUCE", "1") != "0"
        bwd.deferred, bwd_b.deferred = ([] if defer else None), ([] if defer else None)
        self._cur_ws, self._wg_idx = ws, 0
        for bn in self._bn_layers:
            bn.use(ws)
        if not training:     # eval mode (the frozen teacher, evaluation): pack + split only when the parameters changed (pack_if_stale)
            with recording(pack):
                self.pack_all()
        if split_pack:
            with recording(pack_late):
                self.pack_group(1)
        with recording(fwd), K.conv_terms(K.terms_for(getattr(self, "role", "tpg"), "fwd")):
            self._record_fwd(N, training, ws, cut_to=fwd_b if split_pack else None)
        if training:
            with recording(bwd), K.conv_terms(K.terms_for("tpg", "bwd")):
                # the pass is recorded as TWO plans, cut behind the early slab reduce: every gradient from conv3 to the end of the
                # parameter arena (both BiLSTMs, conv6..conv3, bn6 / bn4: 95.6 % of it) is final there, and a data-parallel train step
                # launches that bucket's all-reduce between the two (backward(..., after_early=...)) under the rest of the pass
                self._record_bwd(N, ws, cut_to=bwd_b if os.environ.get("TPGSR_CRNN_BWD_SPLIT", "1") != "0" else None)
                if defer:
                    K.flush_wgrad_reduces()
                K._REC.join()
            with recording(dgp), K.conv_terms(K.terms_for("tpg", "bwd")):   # d gray: only later cascade stages ask for it
                self._record_dgray(N, ws)
        out = dict(fwd=fwd, bwd=bwd, dgray=dgp, pack=pack, ws=ws)
        if len(bwd_b):
            out["bwd_b"] = bwd_b
        if split_pack:
            out["fwd_b"], out["pack_late"] = fwd_b, pack_late
        return out

    def _record_fwd(self, N, training, ws, cut_to=None):
        if training:
            if cut_to is not None:
                self.pack_group(0)       # (group 1: the plan `pack_late`, run before the second half)
            else:
                self.pack_all()
        dims = self._dims()
        cur, loader = K.DynPtr("gray"), {}        # conv0's im2col reads the caller's tensor directly (patched per call)
        for i, conv in enumerate(self.convs):
            if cut_to is not None and i == self.EARLY_CONVS:
                K.continue_in(cut_to)        # everything from here on reads operands of pack group 1
            (h, w), (oh, ow), (ph, pw) = dims[i]
            s = ws(f"s{i}", N * oh * ow, conv.Cout)
            bn = self.bns.get(i)
            part = bn.partial(N * oh * ow)[0] if (bn and training) else None
            fin = None
            if i == 0:
                conv.fwd(N, h, w, cur, ws("col0", N * h * w, 12), s)
            else:
                fin = bn.fin(N * oh * ow, conv.b) if (bn and training) else None      # finalized by the convolution's own launch
                conv.fwd(N, h, w, cur, s, bn_partial=part, bn_fin=fin, **loader)
            if bn and fin is None:
                bn.finalize(N * oh * ow, conv.b, training)
            if i in self.POOLS:
                k, st, pd = self.POOLS[i]
                a = ws(f"a{i}", N * ph * pw, conv.Cout)
                K.pool2d_fwd(s, N, oh, ow, conv.Cout, bn.scale if bn else None, bn.shift if bn else None, "relu", k, st, pd, a)
                cur, loader = a, {}
            else:   # BN(+ReLU) rides on the consumer's loader (conv2, conv4, conv6 are always followed by BN)
                cur, loader = s, (dict(in_act="relu", **bn.loader) if bn else dict(in_act="relu"))
        T = dims[6][1][1]
        self.T = T
        x = cur
        for j, L in enumerate(self.lstm):
            G4 = 4 * L.Hh
            G = ws(f"l{j}_G", N * T, 2 * G4)
            gh = ws(f"l{j}_gh", L.Hh // 32, 2, N, G4)
            Cst = ws(f"l{j}_C", N * T, 2 * L.Hh)
            out = ws(f"l{j}_out", N * T, 2 * L.Hh)
            e = ws(f"l{j}_e", N * T, L.emb.Cout)
            L.fwd(N, T, x, G, gh, Cst, out, e, **(loader if j == 0 else {}))
            x = e
        K.copy(x, K.DynPtr("logits"), N * T * self.nclass)

    def _record_bwd(self, N, ws, cut_to=None):
        t = ws.t
        dims = self._dims()
        T = self.T
        cp = self.lstm[1].emb.Cp if isinstance(self.lstm[1].emb, PaddedLinear) else self.nclass
        de = ws("dlogits", N * T, cp)                 # zero-padded to a multiple of 4 classes (PaddedLinear)
        K.pad_channels(K.DynPtr("dlogits"), N * T, self.nclass, cp, de)
        cnn_loader = dict(in_act="relu", **self.bns[6].loader)
        fz = None                                 # BatchNorm-backward sums already left behind by the producer of `da`
        # side batches (K.side_batch_begin): 1 = both BiLSTMs' weight gradients behind one fork and conv6..conv3's (+ the early slab
        # reduce) behind another; 2 = conv2 / conv1 as well; conv0's stays on its own (it is the tail of the pass)
        # (default 0 since the BiLSTM layers' weight gradients are ONE launch each: with the weight-gradient stream no longer busy with ten
        #  small GEMMs, holding conv6..conv3's back until conv3's data gradient is out only delays them -- C3 6.20 -> 6.05 ms per step)
        sbl = int(os.environ.get("TPGSR_SIDE_BATCH_CRNN", "0"))
        sb = K.side_batch_begin() if sbl >= 1 else False
        for j in (1, 0):
            L = self.lstm[j]
            G4 = 4 * L.Hh
            dout = ws(f"l{j}_dout", N * T, 2 * L.Hh)
            dhc = ws(f"l{j}_dhc", 4 * L.Hh // 32, 2, N, L.Hh)
            dcc = ws(f"l{j}_dcc", N, 2 * L.Hh)
            x = t[f"l{j - 1}_e"] if j == 1 else t["s6"]
            dx = ws(f"l{j}_dx", N * T, L.Cin)
            if j == 0:    # the first BiLSTM's input is relu(bn6(s6)): bn6's backward sums ride on the projection's data gradient
                fz = self.bns[6].fuse_stats(t["s6"], N * T, "relu", 2 * G4)
            L.bwd(N, T, x, t[f"l{j}_G"], t[f"l{j}_C"], t[f"l{j}_out"], de, dout, dhc, dcc, dx, dx_bnb=fz, **({} if j == 1 else cnn_loader))
            de = dx
        da = de                                   # d relu(bn6(s6))
        K.side_batch_end(sb)
        sb = False
        for i in range(6, -1, -1):
            conv, bn = self.convs[i], self.bns.get(i)
            if not sb and (sbl >= 1 and i == 6 or sbl >= 2 and i == 2):
                sb = K.side_batch_begin()
            (h, w), (oh, ow), (ph, pw) = dims[i]
            M = N * oh * ow
            s = t[f"s{i}"]
            ds = ws(f"ds{i}", M, conv.Cout)
            if i in self.POOLS:
                k, st, pd = self.POOLS[i]
                K.pool2d_bwd(s, da, N, oh, ow, conv.Cout, bn.scale if bn else None, bn.shift if bn else None, "relu", k, st, pd, ds)
                if bn:   # (not the case in CRNN: pooled convs have no BN) dz -> BN backward
                    dz = ds
                    ds = ws(f"ds{i}b", M, conv.Cout)
                    bn.backward(dz, None, s, M, "none", ds)
            elif bn:
                bn.backward(da, None, s, M, "relu", ds, fused=fz)
            else:
                K.act_bwd(s, da, M * conv.Cout, "relu", ds)
            # the conv's own input and the loader it was read through
            if i == 0:
                K.side_batch_end(sb)
                conv.wgrad(N, h, w, t["col0"], ds)
                continue
            else:
                pi, pbn = i - 1, self.bns.get(i - 1)
                if pi in self.POOLS:
                    xin, ld = t[f"a{pi}"], {}
                else:
                    xin, ld = t[f"s{pi}"], (dict(in_act="relu", **pbn.loader) if pbn else dict(in_act="relu"))
            conv.wgrad(N, h, w, xin, ds, loader=ld)
            if i == 3 and K.deferring() and os.environ.get("TPGSR_CRNN_EARLY_REDUCE", "1") != "0":
                # the slab reduce of everything recorded so far (both BiLSTMs, conv6..conv3: 92 % of the parameters) goes out now and
                # overlaps with the rest of this backward pass; the reduce at the end of the plan -- on the step's critical tail,
                # nothing is left to hide it -- then only covers conv2..conv0
                K.flush_wgrad_reduces()
                K.side_batch_end(sb)
                sb = False
                if cut_to is not None:
                    K.continue_in(cut_to)
            da = ws(f"da{i - 1}", N * h * w, conv.Cin)
            # conv2 / conv4 are followed by BatchNorm + ReLU and no pooling: the sums of that BatchNorm's backward pass ride on this launch
            fz = pbn.fuse_stats(t[f"s{pi}"], N * h * w, "relu", conv.Cout) if (pbn and pi not in self.POOLS) else None
            conv.dgrad(N, h, w, ds, da, **(dict(bnb=fz) if fz else {}))

    def _record_dgray(self, N, ws):
        h, w = self.IMG_HW
        dcol = ws("dcol0", N * h * w, 12)
        self.convs[0].dgrad(N, h, w, ws.t["ds0"], dcol, K.DynPtr("dgray"))

    # ---- execution ----------------------------------------------------------------------------------------------
    def forward(self, gray: torch.Tensor, training: bool, slot: int = 0, late_stream=None, after_late=None) -> torch.Tensor:
        """gray (N, 1, 32, 100) -> logits [N][T][nclass] (batch-major; the module returns the (T, N, C) view).
        late_stream (training): the stream the operands behind conv2 are packed on, next to conv0..conv2 on the caller's stream (which
        waits for it before conv3); None: everything in order on the caller's stream.  after_late(): called once that packing is enqueued
        (or right away when there is none) -- a train step queues the SR network's prologue behind it on the same stream"""
        if not gray.is_cuda and not K.DRYRUN:
            raise RuntimeError("tpgsr_amd runs on the GPU only (no CPU fallback): move the module and inputs to cuda")
        if gray.dim() != 4 or gray.shape[1] != 1 or tuple(gray.shape[2:]) != self.IMG_HW:
            raise ValueError(f"CRNN expects (N, 1, {self.IMG_HW[0]}, {self.IMG_HW[1]}) input, got {tuple(gray.shape)}")
        self.bind(gray.device)
        N = gray.shape[0]
        pl = self.plans(N, training, slot)
        gray = gray.contiguous().float()            # (N,1,H,W) NCHW with C = 1 is already NHWC
        logits = torch.empty(N, self.T, self.nclass, dtype=F32, device=gray.device)
        fwd = pl["fwd"]
        fwd.set_ptr("gray", gray.data_ptr())
        if "fwd_b" in pl:
            cur, ev = K.current_stream(), None
            if late_stream is not None and late_stream is not cur:
                K.order(late_stream, cur)        # (the previous step's optimiser wrote the parameters on the caller's stream)
                with K.stream_ctx(late_stream):
                    pl["pack_late"].run()
                ev = K.event_record(late_stream)
            else:
                pl["pack_late"].run()
            if after_late is not None:
                after_late()
            fwd.run()
            K.event_wait(cur, ev)
            pl["fwd_b"].set_ptr("logits", logits.data_ptr())
            pl["fwd_b"].run()
        else:
            if after_late is not None:
                after_late()
            fwd.set_ptr("logits", logits.data_ptr())
            if not training:
                self.pack_if_stale(pl["pack"])
            fwd.run()
        if training:
            self.note_packed()
            self._pending_batches += 1
        self._last_gray = gray
        return logits

    #: first parameter (arena order) whose gradient is NOT final when the first backward plan has run: everything from here to the
    #: end of the arena is (both BiLSTMs, conv6..conv3 and their BatchNorms)
    EARLY_FINAL_FROM = "cnn.conv3.weight"

    def early_final_offset(self) -> Optional[int]:
        """offset (floats, in this network's gradient arena) from which on the gradients are final after the FIRST backward plan --
        None when the pass is recorded as one plan (TPGSR_CRNN_EARLY_REDUCE=0 / no deferred reduces)"""
        if "0" in (os.environ.get("TPGSR_CRNN_EARLY_REDUCE", "1"), os.environ.get("TPGSR_DEFER_REDUCE", "1"), os.environ.get("TPGSR_CRNN_BWD_SPLIT", "1")):
            return None
        return self.arena.layout()[0][self.EARLY_FINAL_FROM]

    def backward(self, N, gray: torch.Tensor, dlogits: torch.Tensor, need_dgray: bool = False, slot: int = 0,
                 after_early=None) -> Optional[torch.Tensor]:
        """after_early(): called between the two backward plans, when every gradient from early_final_offset() on is final (on the
        weight-gradient stream + this stream) -- a data-parallel step launches that bucket's all-reduce there"""
        pl = self.plans(N, True, slot)
        self.arena.attach_grads()
        bwd = pl["bwd"]
        dlogits = dlogits.contiguous().float()
        bwd.set_ptr("dlogits", dlogits.data_ptr())
        bwd.run()
        if "bwd_b" in pl:
            if after_early is not None:
                after_early()
            pl["bwd_b"].run()
        if not need_dgray:
            return None
        dgray = torch.empty_like(gray)
        pl["dgray"].set_ptr("dgray", dgray.data_ptr())
        pl["dgray"].run()
        return dgray
