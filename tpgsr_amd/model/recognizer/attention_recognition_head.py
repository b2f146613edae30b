"""Attention GRU decoder (reference: model/recognizer/attention_recognition_head.py): AttentionUnit (:168-218), DecoderUnit (:221-268),
AttentionRecognitionHead.sample (:47-67) = the greedy decode.  Same state_dict layout.  `beam_search` is not provided: the reference's
raises on torch >= 1.5 (integer true division at :111), so there is nothing to pin it against."""
import torch
from torch import nn

from ... import functional as Fh
from ... import kernels as K
from ..nn_params import LinearParams, _NoForward, _uniform


class AttentionUnit(nn.Module):
    def __init__(self, sDim, xDim, attDim):
        super().__init__()
        self.sDim, self.xDim, self.attDim = sDim, xDim, attDim
        self.sEmbed = LinearParams(sDim, attDim)
        self.xEmbed = LinearParams(xDim, attDim)
        self.wEmbed = LinearParams(attDim, 1)


class _EmbeddingParams(_NoForward):
    def __init__(self, num, dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num, dim
        self.weight = nn.Parameter(torch.randn(num, dim))


class _GRUCellParams(_NoForward):
    """nn.GRU(input_size, hidden_size, batch_first=True), one layer, one direction"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        b = 1.0 / hidden_size ** 0.5
        self.weight_ih_l0 = nn.Parameter(_uniform(torch.empty(3 * hidden_size, input_size), b))
        self.weight_hh_l0 = nn.Parameter(_uniform(torch.empty(3 * hidden_size, hidden_size), b))
        self.bias_ih_l0 = nn.Parameter(_uniform(torch.empty(3 * hidden_size), b))
        self.bias_hh_l0 = nn.Parameter(_uniform(torch.empty(3 * hidden_size), b))

    def flatten_parameters(self):
        pass


class DecoderUnit(nn.Module):
    def __init__(self, sDim, xDim, yDim, attDim):
        super().__init__()
        self.sDim, self.xDim, self.yDim, self.attDim, self.emdDim = sDim, xDim, yDim, attDim, attDim
        self.attention_unit = AttentionUnit(sDim, xDim, attDim)
        self.tgt_embedding = _EmbeddingParams(yDim + 1, self.emdDim)        # the last row is <BOS>
        self.gru = _GRUCellParams(xDim + self.emdDim, sDim)
        self.fc = LinearParams(sDim, yDim)


class AttentionRecognitionHead(nn.Module):
    """input: encoder features (N, T, in_planes); sample(): greedy ids (N, max_len_labels) int64 and their softmax scores"""

    def __init__(self, num_classes, in_planes, sDim, attDim, max_len_labels):
        super().__init__()
        self.num_classes, self.in_planes, self.sDim, self.attDim, self.max_len_labels = num_classes, in_planes, sDim, attDim, max_len_labels
        self.decoder = DecoderUnit(sDim=sDim, xDim=in_planes, yDim=num_classes, attDim=attDim)

    def forward(self, x):
        raise RuntimeError("teacher-forced decoding (training) is not part of the evaluation path; use sample()")

    def beam_search(self, x, beam_width, eos):
        raise NotImplementedError("the reference's beam_search raises on torch >= 1.5 (attention_recognition_head.py:111): unpinned, not built")

    def sample(self, x):
        feats = x[0] if isinstance(x, (list, tuple)) else x
        if feats.requires_grad:
            raise RuntimeError("AttentionRecognitionHead.sample is an evaluation path (no gradient)")
        with torch.no_grad():
            d, au = self.decoder, self.decoder.attention_unit
            N, T, D = feats.shape
            dev = feats.device
            feats = feats.contiguous()
            lin = Fh.PackedLinear
            xe, se, gi_l, gh_l, fc = (lin(au.xEmbed.weight, au.xEmbed.bias), lin(au.sEmbed.weight, au.sEmbed.bias),
                                      lin(d.gru.weight_ih_l0, d.gru.bias_ih_l0), lin(d.gru.weight_hh_l0, d.gru.bias_hh_l0),
                                      lin(d.fc.weight, d.fc.bias))
            xproj = xe(feats.reshape(N * T, D))                                   # does not depend on the step (reference recomputes it)
            s = torch.zeros(N, self.sDim, device=dev)
            s2 = torch.empty_like(s)
            y = torch.full((N,), self.num_classes, dtype=torch.int32, device=dev)  # <BOS>
            L = self.max_len_labels
            ids = torch.zeros(N, L, dtype=torch.int32, device=dev)
            scores = torch.zeros(N, L, device=dev)
            alpha, ctx = torch.empty(N, T, device=dev), torch.empty(N, D, device=dev)
            inp = torch.empty(N, self.attDim + D, device=dev)
            wv, bv, emb = au.wEmbed.weight.reshape(-1).contiguous(), au.wEmbed.bias, d.tgt_embedding.weight
            for i in range(L):
                sproj = se(s)
                K.aster_attention(xproj, sproj, wv, bv, feats, N, T, self.attDim, D, alpha, ctx)
                K.embed_concat(y, emb, self.num_classes + 1, self.attDim, ctx, D, N, inp)
                K.gru_cell(gi_l(inp), gh_l(s), s, N, self.sDim, s2)
                s, s2 = s2, s
                K.softmax_max(fc(s), N, self.num_classes, ids, scores, L, i, y)
            return ids.long(), scores
