"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden_aster.py; never by the product path).

CPU restatement (PyTorch functional, fp32) of the reference's ASTER evaluation recognizer on the greedy path
(SURVEY.md section 8 row N2): `parse_aster_data` (interfaces/base.py:844-864) -> STN head + TPS rectification
(model/recognizer/recognizer_builder.py:70-79, stn_head.py:27-95, tps_spatial_transformer.py:54-115) -> ResNet_ASTER encoder with its
two-layer BiLSTM (resnet_aster.py:35-131) -> AttentionRecognitionHead.sample() (attention_recognition_head.py:47-67, DecoderUnit :221-268,
AttentionUnit :168-218), and the string decode of utils/metrics.py:20-70 (`get_string_aster`) over utils/labelmaps.py:6-28.

The reference's own eval forward calls `beam_search`, which raises on torch >= 1.5 (integer true division, attention_recognition_head.py:111);
the greedy `sample()` is the decode that runs, and it is what this file (and the HIP path) pins.  Pinned by make_golden_aster.py, which
imports the genuine reference, asserts equality with every function here and writes tests/golden/aster_eval.npz."""
import string
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import tpgsr_oracle as O

TPS_INPUT, TPS_OUTPUT, N_CTRL, MARGINS = (32, 64), (32, 100), 20, (0.05, 0.05)
LAYERS = [(32, 3, (2, 2)), (64, 4, (2, 2)), (128, 6, (2, 1)), (256, 6, (2, 1)), (512, 3, (2, 1))]


def get_vocabulary(voc_type: str = "all", EOS="EOS", PADDING="PADDING", UNKNOWN="UNKNOWN") -> List[str]:
    """utils/labelmaps.py:6-28"""
    voc = {"digit": string.digits, "lower": string.digits + string.ascii_lowercase,
           "upper": string.digits + string.ascii_letters,
           "all": string.digits + string.ascii_letters + string.punctuation}[voc_type]
    return list(voc) + [EOS, PADDING, UNKNOWN]


def parse_aster_data(imgs: Tensor) -> Tensor:
    """interfaces/base.py:844-864 (fixed-resolution branch): RGB planes, bicubic to 32x128, [0,1] -> [-1,1]"""
    return F.interpolate(imgs[:, :3], (32, 128), mode="bicubic") * 2 - 1


def stn_head(p: Dict[str, Tensor], prefix: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    """model/recognizer/stn_head.py:27-95 in eval mode: 6 x (conv3x3 + BN + ReLU), 2x2 max-pool after the first five"""
    for i in range(6):
        cp = f"{prefix}.stn_convnet.{2 * i}"
        x = F.conv2d(x, p[cp + ".0.weight"], p[cp + ".0.bias"], padding=1)
        x = F.relu(O.batch_norm(p, cp + ".1", x, False))
        if i < 5:
            x = F.max_pool2d(x, 2, 2)
    x = x.reshape(x.shape[0], -1)
    feat = F.relu(O.batch_norm(p, prefix + ".stn_fc1.1", F.linear(x, p[prefix + ".stn_fc1.0.weight"], p[prefix + ".stn_fc1.0.bias"]), False))
    ctrl = F.linear(0.1 * feat, p[prefix + ".stn_fc2.weight"], p[prefix + ".stn_fc2.bias"])
    return feat, ctrl.reshape(-1, N_CTRL, 2)


def aster_block(p, prefix: str, x: Tensor, stride, down: bool) -> Tensor:
    """AsterBlock, resnet_aster.py:35-61: conv1x1(stride) - BN - ReLU - conv3x3 - BN (+ conv1x1(stride) - BN shortcut) - ReLU"""
    out = F.relu(O.batch_norm(p, prefix + ".bn1", F.conv2d(x, p[prefix + ".conv1.weight"], None, stride=stride), False))
    out = O.batch_norm(p, prefix + ".bn2", F.conv2d(out, p[prefix + ".conv2.weight"], None, padding=1), False)
    res = x
    if down:
        res = O.batch_norm(p, prefix + ".downsample.1", F.conv2d(x, p[prefix + ".downsample.0.weight"], None, stride=stride), False)
    return F.relu(out + res)


def lstm2_bidir(p, prefix: str, x: Tensor) -> Tensor:
    """nn.LSTM(512, 256, bidirectional, num_layers=2, batch_first) (resnet_aster.py:88): x (B, T, C) -> (B, T, 512)"""
    y = x.transpose(0, 1)
    for layer in range(2):
        q = {}
        for suf in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                q[f"r.{nm}_l0{suf}"] = p[f"{prefix}.{nm}_l{layer}{suf}"]
        y = O.lstm_bidir_explicit(y, q, "r")
    return y.transpose(0, 1)


def encoder(p, prefix: str, x: Tensor) -> Tensor:
    """ResNet_ASTER.forward (with_lstm), resnet_aster.py:108-131: (N, 3, 32, 100) -> (N, 25, 512)"""
    x = F.relu(O.batch_norm(p, prefix + ".layer0.1", F.conv2d(x, p[prefix + ".layer0.0.weight"], None, padding=1), False))
    inplanes = 32
    for li, (planes, blocks, stride) in enumerate(LAYERS, start=1):
        for b in range(blocks):
            first = b == 0
            x = aster_block(p, f"{prefix}.layer{li}.{b}", x, stride if first else (1, 1), first and (stride != (1, 1) or inplanes != planes))
        inplanes = planes
    feat = x.squeeze(2).transpose(2, 1)
    return lstm2_bidir(p, prefix + ".rnn", feat)


def decoder_step(p, prefix: str, x: Tensor, xproj: Tensor, s: Tensor, y_prev: Tensor) -> Tuple[Tensor, Tensor]:
    """DecoderUnit.forward, attention_recognition_head.py:256-268 (AttentionUnit :196-218 inlined; xProj does not depend on the step)"""
    a = prefix + ".attention_unit"
    sproj = F.linear(s, p[a + ".sEmbed.weight"], p[a + ".sEmbed.bias"])
    v = F.linear(torch.tanh(sproj.unsqueeze(1) + xproj), p[a + ".wEmbed.weight"], p[a + ".wEmbed.bias"]).squeeze(2)
    alpha = F.softmax(v, dim=1)
    context = torch.bmm(alpha.unsqueeze(1), x).squeeze(1)
    inp = torch.cat([p[prefix + ".tgt_embedding.weight"][y_prev], context], 1)
    gi = F.linear(inp, p[prefix + ".gru.weight_ih_l0"], p[prefix + ".gru.bias_ih_l0"])
    gh = F.linear(s, p[prefix + ".gru.weight_hh_l0"], p[prefix + ".gru.bias_hh_l0"])
    H = s.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    s_new = (1 - z) * n + z * s
    return F.linear(s_new, p[prefix + ".fc.weight"], p[prefix + ".fc.bias"]), s_new


def sample(p, prefix: str, x: Tensor, num_classes: int, max_len: int) -> Tuple[Tensor, Tensor]:
    """AttentionRecognitionHead.sample, attention_recognition_head.py:47-67: greedy ids (N, max_len) and their softmax scores"""
    N = x.shape[0]
    a = prefix + ".decoder.attention_unit"
    xproj = F.linear(x, p[a + ".xEmbed.weight"], p[a + ".xEmbed.bias"])
    s = x.new_zeros(N, p[prefix + ".decoder.gru.weight_hh_l0"].shape[1])
    y = torch.full((N,), num_classes, dtype=torch.long)            # the extra embedding row is <BOS>
    ids, scores = [], []
    for _ in range(max_len):
        logits, s = decoder_step(p, prefix + ".decoder", x, xproj, s, y)
        sc, y = F.softmax(logits, dim=1).max(1)
        ids.append(y)
        scores.append(sc)
    return torch.stack(ids, 1), torch.stack(scores, 1)


def aster_greedy(p: Dict[str, Tensor], images: Tensor, num_classes: int, max_len: int) -> Dict[str, Tensor]:
    """RecognizerBuilder (STN_ON) eval path with the greedy decoder: images (N, 3, 32, 128) in [-1, 1]"""
    stn_in = F.interpolate(images, TPS_INPUT, mode="bilinear", align_corners=True)
    _, ctrl = stn_head(p, "stn_head", stn_in)
    rect, _ = O.tps_transform(p, "tps", images, ctrl, TPS_OUTPUT)
    feats = encoder(p, "encoder", rect)
    ids, scores = sample(p, "decoder", feats, num_classes, max_len)
    return {"ctrl": ctrl, "rectified": rect, "feats": feats, "ids": ids, "scores": scores}


def get_string_aster(ids: Tensor, voc: List[str]) -> List[str]:
    """prediction half of utils/metrics.py:20-70: characters up to the first EOS, UNKNOWN skipped"""
    eos, unk = voc.index("EOS"), voc.index("UNKNOWN")
    out = []
    for row in ids.tolist():
        chars = []
        for c in row:
            if c == eos:
                break
            if c != unk:
                chars.append(voc[c])
        out.append("".join(chars))
    return out
