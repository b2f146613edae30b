"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_golden_moran.py; never by the product path).

CPU restatement (PyTorch functional, fp32) of the reference's MORAN evaluation recognizer on the path `--test_model MORAN` takes
(interfaces/super_resolution.py:1389-1396): `parse_moran_data` (interfaces/base.py:608-632) -> MORN rectifier in test mode with one
enhancement pass (model/moran/morn.py:46-79) -> ASRN (model/moran/asrn_res.py:214-259): ResNet (:157-212), two BidirectionalLSTM
(:9-25), bidirectional attention decoder in test mode (Attention :126-155, AttentionCell :39-65) -> arg-max, string decode of
utils/utils_moran.py:79-107 and the cut at '$'.

The evaluation loop calls the model with debug=True, which only ADDS a visualisation (matplotlib / colour / cv2, morn.py:81-137) next
to the same predictions; the numbers pinned here are those of debug=False.  grid_sample runs with the installed torch's default
(align_corners=False), like every other oracle in this directory.  Pinned by make_golden_moran.py, which imports the
genuine reference, asserts equality with every function here and writes tests/golden/moran_eval.npz."""
import string
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import tpgsr_oracle as O

TARGET_HW = (32, 100)
MAX_ITER = 20                 # interfaces/base.py:628: every sample is decoded for 20 steps
BLOCKS = [(32, 2, 3), (64, 2, 4), (128, (2, 1), 6), (256, (2, 1), 6), (512, (2, 1), 3)]      # asrn_res.py:192-196 (c_out, stride, repeat)


def alphabet() -> List[str]:
    """interfaces/base.py:589 / :232-234: digits + lower-case letters + '$' (the end mark)"""
    return list(string.digits + string.ascii_lowercase + "$")


def parse_moran_data(imgs: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """interfaces/base.py:608-632 (fixed-resolution branch): luminance of the bicubic 32x100 resize; dummy targets of 20 x '0'"""
    x = F.interpolate(imgs[:, :3], TARGET_HW, mode="bicubic")
    gray = 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
    B = imgs.shape[0]
    text = torch.zeros(B * MAX_ITER, dtype=torch.long)              # index of '0' in the alphabet
    length = torch.full((B,), MAX_ITER, dtype=torch.long)
    return gray, length, text, text


def base_grid(B: int) -> Tensor:
    """morn.py:27-43: the regular sampling grid, (B, 32, 100, 2) with (x, y) in [-1, 1]"""
    H, W = TARGET_HW
    ys = torch.arange(H, dtype=torch.float64) * 2.0 / (H - 1) - 1
    xs = torch.arange(W, dtype=torch.float64) * 2.0 / (W - 1) - 1
    g = torch.stack([xs.view(1, W).expand(H, W), ys.view(H, 1).expand(H, W)], -1).float()
    return g.unsqueeze(0).expand(B, H, W, 2).contiguous()


def morn_cnn(p, prefix: str, x: Tensor) -> Tensor:
    """morn.py:15-22 in eval mode: pool - 5 x (conv3x3 + BN [+ ReLU]) with pools after the first two"""
    c = prefix + ".cnn"
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(O.batch_norm(p, c + ".2", F.conv2d(x, p[c + ".1.weight"], p[c + ".1.bias"], padding=1), False))
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(O.batch_norm(p, c + ".6", F.conv2d(x, p[c + ".5.weight"], p[c + ".5.bias"], padding=1), False))
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(O.batch_norm(p, c + ".10", F.conv2d(x, p[c + ".9.weight"], p[c + ".9.bias"], padding=1), False))
    x = F.relu(O.batch_norm(p, c + ".13", F.conv2d(x, p[c + ".12.weight"], p[c + ".12.bias"], padding=1), False))
    return O.batch_norm(p, c + ".16", F.conv2d(x, p[c + ".15.weight"], p[c + ".15.bias"], padding=1), False)


def morn_offsets(p, prefix: str, x: Tensor, grid: Tensor) -> Tensor:
    """morn.py:60-66: offset map of the 2x1-pooled positive / negative parts, bilinearly read out on the regular grid -> (B, 32, 100, 1)"""
    off = morn_cnn(p, prefix, x)
    pooled = F.max_pool2d(F.relu(off), 2, 1) - F.max_pool2d(F.relu(-off), 2, 1)
    return F.grid_sample(pooled, grid, align_corners=False).permute(0, 2, 3, 1).contiguous()


def morn(p, prefix: str, x: Tensor, enhance: int = 1) -> Dict[str, Tensor]:
    """MORN.forward(test=True), morn.py:46-79: y-offsets from the image, one enhancement pass from the rectified image"""
    B = x.shape[0]
    grid = base_grid(B)
    gx, gy = grid[..., 0:1], grid[..., 1:2]
    x_small = F.interpolate(x, TARGET_HW, mode="bilinear", align_corners=False)
    og = morn_offsets(p, prefix, x_small, grid)
    rect = F.grid_sample(x, torch.cat([gx, gy + og], 3), align_corners=False)
    for _ in range(enhance):
        og = og + morn_offsets(p, prefix, rect, grid)
        rect = F.grid_sample(x, torch.cat([gx, gy + og], 3), align_corners=False)
    return {"offsets": og, "rectified": rect}


def residual_block(p, prefix: str, x: Tensor, stride, first: bool) -> Tensor:
    """Residual_block, asrn_res.py:157-186: strided 3x3 (first block of a stage) or 1x1 conv - BN - conv3x3 - BN, no ReLU in between;
    shortcut = conv3x3(stride) - BN in the first block; ReLU after the sum"""
    s = (stride, stride) if isinstance(stride, int) else tuple(stride)
    down = first and s[0] > 1
    if down:
        c1 = F.conv2d(x, p[prefix + ".conv1.0.weight"], p[prefix + ".conv1.0.bias"], stride=s, padding=1)
    else:
        c1 = F.conv2d(x, p[prefix + ".conv1.0.weight"], p[prefix + ".conv1.0.bias"])
    c1 = O.batch_norm(p, prefix + ".conv1.1", c1, False)
    c2 = O.batch_norm(p, prefix + ".conv2.1", F.conv2d(c1, p[prefix + ".conv2.0.weight"], p[prefix + ".conv2.0.bias"], padding=1), False)
    res = x
    if down:
        res = O.batch_norm(p, prefix + ".downsample.1",
                           F.conv2d(x, p[prefix + ".downsample.0.weight"], p[prefix + ".downsample.0.bias"], stride=s, padding=1), False)
    return F.relu(res + c2)


def resnet(p, prefix: str, x: Tensor) -> Tensor:
    """ResNet.forward, asrn_res.py:188-212: (B, 1, 32, 100) -> (B, 512, 1, 25); block0 is conv + BN WITHOUT an activation"""
    x = O.batch_norm(p, prefix + ".block0.1", F.conv2d(x, p[prefix + ".block0.0.weight"], p[prefix + ".block0.0.bias"], padding=1), False)
    for bi, (_c, stride, repeat) in enumerate(BLOCKS, start=1):
        for r in range(repeat):
            x = residual_block(p, f"{prefix}.block{bi}.{r}", x, stride if r == 0 else 1, r == 0)
    return x


def bilstm(p, prefix: str, x: Tensor) -> Tensor:
    """BidirectionalLSTM, asrn_res.py:9-25 (eval: the dropout of the one-layer nn.LSTM never applies): x (T, B, C) -> (T, B, nOut)"""
    q = {f"r.{nm}_l0{suf}": p[f"{prefix}.rnn.{nm}_l0{suf}"] for suf in ("", "_reverse") for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    rec = O.lstm_bidir_explicit(x, q, "r")
    return F.linear(rec, p[prefix + ".embedding.weight"], p[prefix + ".embedding.bias"])


def attention_step(p, prefix: str, feats: Tensor, feats_proj: Tensor, hidden: Tensor, emb: Tensor) -> Tensor:
    """AttentionCell.forward(test=True), asrn_res.py:39-65; feats (T, B, C); feats_proj = i2h(feats) does not depend on the step"""
    c = prefix + ".attention_cell"
    hp = F.linear(hidden, p[c + ".h2h.weight"], p[c + ".h2h.bias"])
    e = F.linear(torch.tanh(feats_proj + hp.unsqueeze(0)), p[c + ".score.weight"]).squeeze(2)          # (T, B)
    alpha = F.softmax(e, 0)
    context = (feats * alpha.unsqueeze(2)).sum(0)
    inp = torch.cat([context, emb], 1)
    gi = F.linear(inp, p[c + ".rnn.weight_ih"], p[c + ".rnn.bias_ih"])
    gh = F.linear(hidden, p[c + ".rnn.weight_hh"], p[c + ".rnn.bias_hh"])
    H = hidden.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * hidden


def attention_test(p, prefix: str, feats: Tensor, length: Tensor) -> Tensor:
    """Attention.forward(test=True), asrn_res.py:126-155: greedy feedback (arg-max + 1 indexes the next embedding row), the class
    scores of the first length[b] steps of every sample, concatenated sample after sample -> (sum(length), nclass)"""
    T, B, C = feats.shape
    steps = int(length.max())
    fp = F.linear(feats, p[prefix + ".attention_cell.i2h.weight"])
    hidden = feats.new_zeros(B, p[prefix + ".attention_cell.rnn.weight_hh"].shape[1])
    tgt = torch.zeros(B, dtype=torch.long)
    out = []
    for _ in range(steps):
        hidden = attention_step(p, prefix, feats, fp, hidden, p[prefix + ".char_embeddings"][tgt])
        logits = F.linear(hidden, p[prefix + ".generator.weight"], p[prefix + ".generator.bias"])
        out.append(logits)
        tgt = logits.argmax(1) + 1
    probs = torch.stack(out, 1)                                     # (B, steps, nclass)
    return torch.cat([probs[b, :int(length[b])] for b in range(B)], 0)


def asrn(p, prefix: str, x: Tensor, length: Tensor) -> Dict[str, Tensor]:
    """ASRN.forward(test=True) with BidirDecoder, asrn_res.py:241-256"""
    conv = resnet(p, prefix + ".cnn", x)
    assert conv.shape[2] == 1, "the height of conv must be 1"
    seq = conv.squeeze(2).permute(2, 0, 1).contiguous()             # (W, B, C)
    rnn = bilstm(p, prefix + ".rnn.1", bilstm(p, prefix + ".rnn.0", seq))
    return {"conv": conv, "rnn": rnn, "l2r": attention_test(p, prefix + ".attentionL2R", rnn, length),
            "r2l": attention_test(p, prefix + ".attentionR2L", rnn, length)}


def moran(p: Dict[str, Tensor], x: Tensor, length: Tensor) -> Dict[str, Tensor]:
    """MORAN.forward(test=True), moran.py:14-22"""
    m = morn(p, "MORN", x)
    out = asrn(p, "ASRN", m["rectified"], length)
    out.update(m)
    return out


def decode(ids: Tensor, length: Tensor) -> List[str]:
    """strLabelConverterForAttention.decode (utils/utils_moran.py:79-107): the alphabet characters of every sample's ids"""
    abc = alphabet()
    out, i = [], 0
    for n in length.tolist():
        out.append("".join(abc[c] for c in ids[i:i + n].tolist()))
        i += n
    return out


def get_string_moran(l2r_logits: Tensor, length: Tensor) -> List[str]:
    """interfaces/super_resolution.py:1393-1396: arg-max of the left-to-right decoder, decoded, cut at the first '$'"""
    return [s.split("$")[0] for s in decode(l2r_logits.argmax(1), length)]
