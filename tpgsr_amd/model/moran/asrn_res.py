"""ASRN, the attention-based sequence recognition network of MORAN (reference: model/moran/asrn_res.py): ResNet (:157-212) ->
two BidirectionalLSTM (:9-25) -> attention GRU decoder(s) in TEST mode (Attention :126-155, AttentionCell :39-65: greedy feedback,
arg-max + 1 selects the next character embedding).  Same state_dict keys / shapes / init.  Evaluation only: the teacher-forced
training branch (with fracPickup) is not part of the evaluation path.  Everything runs on the HIP kernels (tpgsr_amd/functional.py,
csrc/aster.hip for the decoder steps)."""
import math

import torch
from torch import nn

from ... import functional as Fh
from ... import kernels as K
from ..nn_params import BatchNormParams, Conv2dParams, LinearParams, LSTMParams, _NoForward, _uniform


class BidirectionalLSTM(nn.Module):
    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = LSTMParams(nIn, nHidden)            # nn.LSTM(nIn, nHidden, bidirectional=True, dropout=0.3): one layer, the dropout never applies
        self.embedding = LinearParams(nHidden * 2, nOut)

    def forward(self, x):
        """x (N, T, C) batch-first here (the reference is sequence-first) -> (N, T, nOut)"""
        r = self.rnn
        rec = Fh.bilstm_eval(x, *[getattr(r, nm + suf) for suf in ("", "_reverse") for nm in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")])
        N, T, H2 = rec.shape
        return Fh.PackedLinear(self.embedding.weight, self.embedding.bias)(rec.reshape(N * T, H2)).reshape(N, T, -1)


class _GRUCellParams(_NoForward):
    """nn.GRUCell(input_size, hidden_size)"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        b = 1.0 / math.sqrt(hidden_size)
        self.weight_ih = nn.Parameter(_uniform(torch.empty(3 * hidden_size, input_size), b))
        self.weight_hh = nn.Parameter(_uniform(torch.empty(3 * hidden_size, hidden_size), b))
        self.bias_ih = nn.Parameter(_uniform(torch.empty(3 * hidden_size), b))
        self.bias_hh = nn.Parameter(_uniform(torch.empty(3 * hidden_size), b))


class _LinearNoBias(_NoForward):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(_uniform(torch.empty(out_features, in_features), 1.0 / math.sqrt(in_features)))


class AttentionCell(nn.Module):
    def __init__(self, input_size, hidden_size, num_embeddings=128, CUDA=True):
        super().__init__()
        self.i2h = _LinearNoBias(input_size, hidden_size)
        self.h2h = LinearParams(hidden_size, hidden_size)
        self.score = _LinearNoBias(hidden_size, 1)
        self.rnn = _GRUCellParams(input_size + num_embeddings, hidden_size)
        self.hidden_size, self.input_size, self.num_embeddings = hidden_size, input_size, num_embeddings
        self.fracPickup = _NoForward()                 # training-time attention jitter (fracPickup.py); no parameters


class Attention(nn.Module):
    def __init__(self, input_size, hidden_size, num_classes, num_embeddings=128, CUDA=True):
        super().__init__()
        self.attention_cell = AttentionCell(input_size, hidden_size, num_embeddings, CUDA=CUDA)
        self.input_size, self.hidden_size = input_size, hidden_size
        self.generator = LinearParams(hidden_size, num_classes)
        self.char_embeddings = nn.Parameter(torch.randn(num_classes + 1, num_embeddings))
        self.num_embeddings, self.num_classes = num_embeddings, num_classes

    def forward(self, feats, text_length, text, test=False):
        """feats (N, T, C) batch-first here; test mode only: (sum(text_length), num_classes) class scores, sample after sample"""
        if not test or self.training:
            raise RuntimeError("Attention is an evaluation decoder here (teacher forcing is a training path); call .eval() and pass test=True")
        with torch.no_grad():
            cell = self.attention_cell
            N, T, C = feats.shape
            if C != self.input_size or N != text_length.numel():
                raise ValueError(f"feats {tuple(feats.shape)} vs input_size {self.input_size} / {text_length.numel()} lengths")
            lens = [int(v) for v in text_length.tolist()]
            steps, dev, Hd, E = max(lens), feats.device, self.hidden_size, self.num_embeddings
            feats = feats.contiguous()
            lin = Fh.PackedLinear
            w_ih = cell.rnn.weight_ih          # columns: [context (C) | embedding (E)]; the concat kernel writes [embedding | context]
            i2h, h2h = lin(cell.i2h.weight), lin(cell.h2h.weight, cell.h2h.bias)
            gi_l = lin(torch.cat([w_ih[:, C:], w_ih[:, :C]], 1).contiguous(), cell.rnn.bias_ih)
            gh_l, gen = lin(cell.rnn.weight_hh, cell.rnn.bias_hh), lin(self.generator.weight, self.generator.bias)
            fproj = i2h(feats.reshape(N * T, C))                                  # does not depend on the step
            wv, bv = cell.score.weight.reshape(-1).contiguous(), torch.zeros(1, device=dev)
            hidden, hidden2 = torch.zeros(N, Hd, device=dev), torch.empty(N, Hd, device=dev)
            tgt = torch.zeros(N, dtype=torch.int32, device=dev)                   # row 0 of the embedding table starts every sequence
            ids, sc = torch.zeros(N, steps, dtype=torch.int32, device=dev), torch.zeros(N, steps, device=dev)
            alpha, ctx, inp = torch.empty(N, T, device=dev), torch.empty(N, C, device=dev), torch.empty(N, E + C, device=dev)
            logits = torch.empty(steps, N, self.num_classes, device=dev)
            emb_all = self.char_embeddings.detach().contiguous()
            emb_next = emb_all[1:]                                                # arg-max c feeds row c + 1 (asrn_res.py:141-142)
            for i in range(steps):
                K.aster_attention(fproj, h2h(hidden), wv, bv, feats, N, T, Hd, C, alpha, ctx)
                if i == 0:
                    K.embed_concat(tgt, emb_all, self.num_classes + 1, E, ctx, C, N, inp)
                else:
                    K.embed_concat(tgt, emb_next, self.num_classes, E, ctx, C, N, inp)
                K.gru_cell(gi_l(inp), gh_l(hidden), hidden, N, Hd, hidden2)
                hidden, hidden2 = hidden2, hidden
                gen(hidden, out=logits[i])
                K.softmax_max(logits[i], N, self.num_classes, ids, sc, steps, i, tgt)
            probs = logits.permute(1, 0, 2)                                       # (N, steps, classes) view
            return torch.cat([probs[b, :lens[b]] for b in range(N)], 0).contiguous()


class Residual_block(nn.Module):
    def __init__(self, c_in, c_out, stride):
        super().__init__()
        s = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.stride = s
        self.downsample = None
        flag = s[0] > 1
        if flag:
            self.downsample = nn.Sequential(Conv2dParams(c_in, c_out, 3, padding=1), BatchNormParams(c_out, momentum=0.01))
            self.conv1 = nn.Sequential(Conv2dParams(c_in, c_out, 3, padding=1), BatchNormParams(c_out, momentum=0.01))
        else:
            self.conv1 = nn.Sequential(Conv2dParams(c_in, c_out, 1, padding=0), BatchNormParams(c_out, momentum=0.01))
        self.conv2 = nn.Sequential(Conv2dParams(c_out, c_out, 3, padding=1), BatchNormParams(c_out, momentum=0.01))
        self.relu = _NoForward()

    def _conv(self, seq, x, strided):
        c = seq[0]
        y = Fh.conv2d_strided(x, c.weight, c.bias, self.stride, c.padding) if strided else c(x)
        return seq[1](y)

    def forward(self, x):
        """x NHWC"""
        strided = self.downsample is not None
        c2 = self._conv(self.conv2, self._conv(self.conv1, x, strided), False)
        res = self._conv(self.downsample, x, True) if strided else x
        return Fh.relu(Fh.add(res, c2))


class ResNet(nn.Module):
    def __init__(self, c_in):
        super().__init__()
        self.block0 = nn.Sequential(Conv2dParams(c_in, 32, 3, padding=1), BatchNormParams(32, momentum=0.01))
        self.block1 = self._make_layer(32, 32, 2, 3)
        self.block2 = self._make_layer(32, 64, 2, 4)
        self.block3 = self._make_layer(64, 128, (2, 1), 6)
        self.block4 = self._make_layer(128, 256, (2, 1), 6)
        self.block5 = self._make_layer(256, 512, (2, 1), 3)

    def _make_layer(self, c_in, c_out, stride, repeat=3):
        layers = [Residual_block(c_in, c_out, stride)]
        for _ in range(repeat - 1):
            layers.append(Residual_block(c_out, c_out, 1))
        return nn.Sequential(*layers)

    def forward(self, x):
        """x NHWC -> NHWC"""
        h = self.block0[1](self.block0[0](x))
        for blk in (self.block1, self.block2, self.block3, self.block4, self.block5):
            for b in blk:
                h = b(h)
        return h


class ASRN(nn.Module):
    def __init__(self, imgH, nc, nclass, nh, BidirDecoder=False, CUDA=True):
        super().__init__()
        assert imgH % 16 == 0, "imgH must be a multiple of 16"
        self.cnn = ResNet(nc)
        self.rnn = nn.Sequential(BidirectionalLSTM(512, nh, nh), BidirectionalLSTM(nh, nh, nh))
        self.BidirDecoder = BidirDecoder
        if BidirDecoder:
            self.attentionL2R = Attention(nh, nh, nclass, 256, CUDA=CUDA)
            self.attentionR2L = Attention(nh, nh, nclass, 256, CUDA=CUDA)
        else:
            self.attention = Attention(nh, nh, nclass, 256, CUDA=CUDA)
        with torch.no_grad():      # asrn_res.py:234-239: kaiming_normal(fan_out) for convs, BN (1, 0)
            for m in self.modules():
                if isinstance(m, Conv2dParams):
                    m.weight.normal_(0, math.sqrt(2.0 / (m.out_channels * m.kernel_size[0] * m.kernel_size[1])))

    def features(self, x):
        """x (N, nc, 32, W) NCHW -> (conv features NHWC (N, 1, W / 4, 512), recurrent features (N, T, nh))"""
        if self.training:
            raise RuntimeError("ASRN is an evaluation recognizer here: call .eval()")
        with torch.no_grad():
            conv = self.cnn(Fh.to_nhwc(x))
            N, h, w, c = conv.shape
            if h != 1:
                raise ValueError("the height of conv must be 1")
            seq = conv.reshape(N, w, c)
            return conv, self.rnn[1](self.rnn[0](seq))

    def forward(self, input, length, text, text_rev, test=False):
        if self.training or not test:
            raise RuntimeError("ASRN is an evaluation recognizer here (interfaces/base.py:603-605 loads MORAN frozen): call .eval() and pass test=True")
        with torch.no_grad():
            _, rnn = self.features(input)
            if self.BidirDecoder:
                return self.attentionL2R(rnn, length, text, test), self.attentionR2L(rnn, length, text_rev, test)
            return self.attention(rnn, length, text, test)
