"""CPU oracle for the TPGSR-TSRN hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 *restatement* (functional style, written
from the behaviour of the reference, not copied from it) of the path named in
BASELINE.json / SURVEY.md section 8.  It is the checker for the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * the product package ``tpgsr_amd`` never imports it and has no CPU fallback.

Parity pin: every function here is checked (<= 1e-5 abs, see
``tests/golden/make_golden.py``) against the genuine reference imported in the
build container, and the outputs are committed as fixtures under
``tests/golden/`` (``tests/test_oracle_golden.py`` re-checks them everywhere).

Every function cites the reference file:line whose behaviour it restates.
All math bottoms out in ATen (conv2d, grid_sample, interpolate) exactly as the
reference does; the recurrent cells are additionally written out explicitly
(``gru_bidir_explicit`` / ``lstm_bidir_explicit``) so the gate equations the HIP
kernels implement are stated in one place.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = "OrderedDict[str, Tensor]"

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ---------------------------------------------------------------------------
# state_dict specifications (key order + shapes of the reference modules)
# ---------------------------------------------------------------------------
def _bn_spec(prefix: str, c: int):
    return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
            (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
            (prefix + ".num_batches_tracked", (), "bn_nbt")]


def _conv_spec(prefix: str, co: int, ci: int, kh: int, kw: int, bias=True):
    s = [(prefix + ".weight", (co, ci, kh, kw), "w")]
    if bias:
        s.append((prefix + ".bias", (co,), "b"))
    return s


def _gru_spec(prefix: str, cin: int, hid: int):
    s = []
    for suf in ("", "_reverse"):
        s += [(prefix + ".weight_ih_l0" + suf, (3 * hid, cin), "w"),
              (prefix + ".weight_hh_l0" + suf, (3 * hid, hid), "w"),
              (prefix + ".bias_ih_l0" + suf, (3 * hid,), "b"),
              (prefix + ".bias_hh_l0" + suf, (3 * hid,), "b")]
    return s


def _lstm_spec(prefix: str, cin: int, hid: int):
    s = []
    for suf in ("", "_reverse"):
        s += [(prefix + ".weight_ih_l0" + suf, (4 * hid, cin), "w"),
              (prefix + ".weight_hh_l0" + suf, (4 * hid, hid), "w"),
              (prefix + ".bias_ih_l0" + suf, (4 * hid,), "b"),
              (prefix + ".bias_hh_l0" + suf, (4 * hid,), "b")]
    return s


def _gru_block_spec(prefix: str, cin: int, cout: int):
    # GruBlock: model/tsrn.py:491-497  (conv1 1x1 then bidirectional GRU(cout, cout/2))
    return _conv_spec(prefix + ".conv1", cout, cin, 1, 1) + _gru_spec(prefix + ".gru", cout, cout // 2)


def _rrb_spec(prefix: str, c: int, text_c: int = 0):
    # RecurrentResidualBlock(TL): model/tsrn.py:373-382 / :397-407 (registration order)
    return (_conv_spec(prefix + ".conv1", c, c, 3, 3) + _bn_spec(prefix + ".bn1", c)
            + _gru_block_spec(prefix + ".gru1", c + text_c, c)
            + _conv_spec(prefix + ".conv2", c, c, 3, 3) + _bn_spec(prefix + ".bn2", c)
            + _gru_block_spec(prefix + ".gru2", c, c))


def _stn_spec(prefix: str, in_planes: int, n_ctrl: int):
    # STNHead: model/stn_head.py:33-53
    chans = [(in_planes, 32), (32, 64), (64, 128), (128, 256), (256, 256), (256, 256)]
    s = []
    for i, (ci, co) in enumerate(chans):
        s += _conv_spec(f"{prefix}.stn_convnet.{2 * i}.0", co, ci, 3, 3)
        s += _bn_spec(f"{prefix}.stn_convnet.{2 * i}.1", co)
    s += [(prefix + ".stn_fc1.0.weight", (512, 512), "w"), (prefix + ".stn_fc1.0.bias", (512,), "b")]
    s += _bn_spec(prefix + ".stn_fc1.1", 512)
    s += [(prefix + ".stn_fc2.weight", (2 * n_ctrl, 512), "stn_fc2_w"),
          (prefix + ".stn_fc2.bias", (2 * n_ctrl,), "stn_fc2_b")]
    return s


def _tps_spec(prefix: str, h: int, w: int, n_ctrl: int):
    # TPSSpatialTransformer buffers: model/tps_spatial_transformer.py:92-95
    return [(prefix + ".inverse_kernel", (n_ctrl + 3, n_ctrl + 3), "tps"),
            (prefix + ".padding_matrix", (3, 2), "tps"),
            (prefix + ".target_coordinate_repr", (h * w, n_ctrl + 3), "tps"),
            (prefix + ".target_control_points", (n_ctrl, 2), "tps")]


def tsrn_spec(scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32,
              text_prior=False, text_emb=37, out_text_channels=32):
    """Ordered (key, shape, kind) list == reference ``TSRN(...).state_dict()`` (model/tsrn.py:18-60)
    or, with ``text_prior=True``, ``TSRN_TL(...).state_dict()`` (model/tsrn.py:111-176)."""
    in_planes = 4 if mask else 3
    c = 2 * hidden_units
    n_up = int(math.log(scale_factor, 2))
    s = _conv_spec("block1.0", c, in_planes, 9, 9) + [("block1.1.weight", (1,), "prelu")]
    for i in range(srb_nums):
        s += _rrb_spec(f"block{i + 2}", c, out_text_channels if text_prior else 0)
    if text_prior:
        # InfoGen: model/tsrn.py:81-98 (ConvTranspose2d weights are (Cin, Cout, kh, kw), no bias)
        tc = [(text_emb, 512), (512, 128), (128, 64), (64, out_text_channels)]
        for i, (ci, co) in enumerate(tc):
            s += [(f"infoGen.tconv{i + 1}.weight", (ci, co, 3, 3), "w")]
            s += _bn_spec(f"infoGen.bn{i + 1}", co)
    s += _conv_spec(f"block{srb_nums + 2}.0", c, c, 3, 3) + _bn_spec(f"block{srb_nums + 2}.1", c)
    for u in range(n_up):
        s += _conv_spec(f"block{srb_nums + 3}.{u}.conv", c * 4, c, 3, 3)
    s += _conv_spec(f"block{srb_nums + 3}.{n_up}", in_planes, c, 9, 9)
    if STN:
        s += _tps_spec("tps", height // scale_factor, width // scale_factor, 20)
        s += _stn_spec("stn_head", in_planes, 20)
    return s


def crnn_spec(imgH=32, nc=1, nclass=37, nh=256):
    """Ordered spec == reference ``CRNN(32,1,37,256).state_dict()`` (model/crnn/crnn.py:29-72)."""
    nm = [64, 128, 256, 256, 512, 512, 512]
    ks = [3, 3, 3, 3, 3, 3, 2]
    bn = {2, 4, 6}
    s = []
    for i in range(7):
        ci = nc if i == 0 else nm[i - 1]
        s += _conv_spec(f"cnn.conv{i}", nm[i], ci, ks[i], ks[i])
        if i in bn:
            s += _bn_spec(f"cnn.batchnorm{i}", nm[i])
    s += _lstm_spec("rnn.0.rnn", 512, nh)
    s += [("rnn.0.embedding.weight", (nh, 2 * nh), "w"), ("rnn.0.embedding.bias", (nh,), "b")]
    s += _lstm_spec("rnn.1.rnn", nh, nh)
    s += [("rnn.1.embedding.weight", (nclass, 2 * nh), "w"), ("rnn.1.embedding.bias", (nclass,), "b")]
    return s


def srcnn_spec(in_planes=3):
    """SRCNN(STN=False): model/srcnn.py:109-117."""
    return (_conv_spec("conv1", 64, in_planes, 9, 9) + _conv_spec("conv2", 32, 64, 1, 1)
            + _conv_spec("conv3", in_planes, 32, 5, 5))


# ---------------------------------------------------------------------------
# TPS constants (model/tps_spatial_transformer.py:22-95)
# ---------------------------------------------------------------------------
def tps_target_control_points(n_ctrl: int, margins: Tuple[float, float]) -> Tensor:
    """build_output_control_points, tps_spatial_transformer.py:38-50 (float64 linspace -> fp32)."""
    mx, my = margins
    k = n_ctrl // 2
    xs = np.linspace(mx, 1.0 - mx, k)
    top = np.stack([xs, np.full(k, my)], axis=1)
    bot = np.stack([xs, np.full(k, 1.0 - my)], axis=1)
    return torch.tensor(np.concatenate([top, bot], 0), dtype=torch.float32)


def tps_partial_repr(pts: Tensor, ctrl: Tensor) -> Tensor:
    """phi(d2) = 0.5*d2*log(d2), NaN -> 0 (tps_spatial_transformer.py:22-34)."""
    d = pts[:, None, :] - ctrl[None, :, :]
    d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    r = 0.5 * d2 * torch.log(d2)
    return torch.where(torch.isnan(r), torch.zeros_like(r), r)


def tps_buffers(h: int, w: int, n_ctrl: int = 20, margins=(0.05, 0.05)) -> Dict[str, Tensor]:
    """The four registered buffers (tps_spatial_transformer.py:64-95), fp32 arithmetic like the reference."""
    tcp = tps_target_control_points(n_ctrl, margins)
    n = n_ctrl
    fk = torch.zeros(n + 3, n + 3)
    fk[:n, :n] = tps_partial_repr(tcp, tcp)
    fk[:n, n] = 1
    fk[n, :n] = 1
    fk[:n, n + 1:] = tcp
    fk[n + 1:, :n] = tcp.t()
    inv = torch.inverse(fk)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                            indexing="ij")
    coord = torch.stack([xs.reshape(-1) / (w - 1), ys.reshape(-1) / (h - 1)], 1)  # (x, y)
    rep = torch.cat([tps_partial_repr(coord, tcp), torch.ones(h * w, 1), coord], 1)
    return {"inverse_kernel": inv, "padding_matrix": torch.zeros(3, 2),
            "target_coordinate_repr": rep, "target_control_points": tcp}


def stn_identity_ctrl_points(n_ctrl: int = 20, margin: float = 0.01) -> Tensor:
    """init_stn bias, model/stn_head.py:73-90."""
    return tps_target_control_points(n_ctrl, (margin, margin)).reshape(-1)


# ---------------------------------------------------------------------------
# weights-by-recipe (SURVEY 8c golden-vector policy)
# ---------------------------------------------------------------------------
def recipe_state_dict(spec, seed: int, tps_hw: Optional[Tuple[int, int]] = None) -> "OrderedDict[str, Tensor]":
    """Deterministic pseudo-trained weights: every tensor is filled in key order from
    ``numpy.random.default_rng(seed)`` with a per-kind scale.  Both the reference (in the golden
    generator) and the build load exactly these tensors, so no weight files are shipped."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, Tensor]" = OrderedDict()
    tps = None
    for name, shape, kind in spec:
        if kind == "w":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            if "tconv" in name:  # ConvTranspose2d weight is (Cin, Cout, kh, kw)
                fan_in = shape[0] * shape[2] * shape[3] // 2
            a = rng.standard_normal(shape) * (1.0 / math.sqrt(max(fan_in, 1)))
        elif kind == "b":
            a = rng.standard_normal(shape) * 0.1
        elif kind == "prelu":
            a = np.full(shape, 0.25) + rng.standard_normal(shape) * 0.02
        elif kind == "bn_w":
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "bn_b":
            a = 0.1 * rng.standard_normal(shape)
        elif kind == "bn_rm":
            a = 0.1 * rng.standard_normal(shape)
        elif kind == "bn_rv":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[name] = torch.zeros((), dtype=torch.long)
            continue
        elif kind == "stn_fc2_w":
            a = rng.standard_normal(shape) * 0.02
        elif kind == "stn_fc2_b":
            a = stn_identity_ctrl_points(shape[0] // 2).numpy() + rng.standard_normal(shape) * 0.03
        elif kind == "tps":
            if tps is None:
                assert tps_hw is not None
                tps = tps_buffers(tps_hw[0], tps_hw[1])
            sd[name] = tps[name.split(".")[-1]].clone()
            continue
        else:
            raise KeyError(kind)
        sd[name] = torch.tensor(np.asarray(a), dtype=torch.float32).reshape(shape)
    return sd


def as_params(sd, requires_grad=True) -> "OrderedDict[str, Tensor]":
    """Clone a state_dict into leaf tensors (float params get requires_grad)."""
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        is_param = t.is_floating_point() and not any(
            s in k for s in ("running_mean", "running_var", "inverse_kernel", "padding_matrix",
                             "target_coordinate_repr", "target_control_points"))
        if is_param and requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def trainable_keys(p) -> List[str]:
    return [k for k, v in p.items() if v.requires_grad]


# ---------------------------------------------------------------------------
# primitive blocks
# ---------------------------------------------------------------------------
def mish(x: Tensor) -> Tensor:
    """x * tanh(softplus(x)) -- model/tsrn.py:480-488."""
    return x * torch.tanh(F.softplus(x))


def batch_norm(p, prefix: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm{1,2}d semantics: batch stats (biased var) in training + running-stat update with
    momentum 0.1 and unbiased var; running stats in eval."""
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if training and (prefix + ".num_batches_tracked") in p:
        p[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)


def gru_bidir_explicit(x: Tensor, p, prefix: str) -> Tensor:
    """Bidirectional single-layer GRU, batch_first, h0 = 0 (nn.GRU as used at model/tsrn.py:496).
    x: (B, T, C) -> (B, T, 2H).  Gate order (r, z, n):
        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr);  z likewise
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn));  h' = (1 - z) * n + z * h
    """
    B, T, _ = x.shape
    outs = []
    for suf, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        w_ih, w_hh = p[prefix + ".weight_ih_l0" + suf], p[prefix + ".weight_hh_l0" + suf]
        b_ih, b_hh = p[prefix + ".bias_ih_l0" + suf], p[prefix + ".bias_hh_l0" + suf]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        gi_all = x @ w_ih.t() + b_ih
        ys = [None] * T
        for t in order:
            gi = gi_all[:, t]
            gh = h @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            ys[t] = h
        outs.append(torch.stack(ys, 1))
    return torch.cat(outs, 2)


def _rnn_flat_weights(p, prefix: str):
    w = []
    for suf in ("", "_reverse"):
        w += [p[prefix + ".weight_ih_l0" + suf], p[prefix + ".weight_hh_l0" + suf],
              p[prefix + ".bias_ih_l0" + suf], p[prefix + ".bias_hh_l0" + suf]]
    return w


def gru_bidir(x: Tensor, p, prefix: str, explicit: bool = False) -> Tensor:
    """Same as gru_bidir_explicit, through ATen's fused CPU GRU (what nn.GRU calls) unless explicit."""
    if explicit:
        return gru_bidir_explicit(x, p, prefix)
    H = p[prefix + ".weight_hh_l0"].shape[1]
    h0 = x.new_zeros(2, x.shape[0], H)
    y, _ = torch._VF.gru(x, h0, _rnn_flat_weights(p, prefix), True, 1, 0.0, False, True, True)
    return y


def lstm_bidir_explicit(x: Tensor, p, prefix: str) -> Tensor:
    """Bidirectional single-layer LSTM, seq-first, zero initial state (nn.LSTM at model/crnn/crnn.py:10).
    x: (T, B, C) -> (T, B, 2H).  Gate order (i, f, g, o):  c' = f*c + i*g;  h' = o * tanh(c')."""
    T, B, _ = x.shape
    outs = []
    for suf, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        w_ih, w_hh = p[prefix + ".weight_ih_l0" + suf], p[prefix + ".weight_hh_l0" + suf]
        b_ih, b_hh = p[prefix + ".bias_ih_l0" + suf], p[prefix + ".bias_hh_l0" + suf]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        g_all = x @ w_ih.t() + b_ih
        ys = [None] * T
        for t in order:
            g = g_all[t] + h @ w_hh.t() + b_hh
            i = torch.sigmoid(g[:, :H])
            f = torch.sigmoid(g[:, H:2 * H])
            gg = torch.tanh(g[:, 2 * H:3 * H])
            o = torch.sigmoid(g[:, 3 * H:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
            ys[t] = h
        outs.append(torch.stack(ys, 0))
    return torch.cat(outs, 2)


def lstm_bidir(x: Tensor, p, prefix: str, explicit: bool = False) -> Tensor:
    if explicit:
        return lstm_bidir_explicit(x, p, prefix)
    H = p[prefix + ".weight_hh_l0"].shape[1]
    z = x.new_zeros(2, x.shape[1], H)
    y, _, _ = torch._VF.lstm(x, (z, z), _rnn_flat_weights(p, prefix), True, 1, 0.0, False, True, False)
    return y


def gru_block(p, prefix: str, x: Tensor, explicit_rnn=False) -> Tensor:
    """GruBlock.forward, model/tsrn.py:498-508: 1x1 conv, then a BiGRU over the LAST spatial axis
    (every (n, row) pair is one sequence of length W)."""
    x = F.conv2d(x, p[prefix + ".conv1.weight"], p[prefix + ".conv1.bias"])
    n, c, h, w = x.shape
    seq = x.permute(0, 2, 3, 1).reshape(n * h, w, c)
    y = gru_bidir(seq, p, prefix + ".gru", explicit_rnn)
    return y.reshape(n, h, w, c).permute(0, 3, 1, 2)


def recurrent_residual_block(p, prefix: str, x: Tensor, training: bool, text_emb: Optional[Tensor] = None,
                             explicit_rnn=False) -> Tensor:
    """RecurrentResidualBlock.forward (model/tsrn.py:384-394) and, with text_emb,
    RecurrentResidualBlockTL.forward (:411-426)."""
    r = F.conv2d(x, p[prefix + ".conv1.weight"], p[prefix + ".conv1.bias"], padding=1)
    r = mish(batch_norm(p, prefix + ".bn1", r, training))
    r = F.conv2d(r, p[prefix + ".conv2.weight"], p[prefix + ".conv2.bias"], padding=1)
    r = batch_norm(p, prefix + ".bn2", r, training)
    if text_emb is not None:
        r = torch.cat([r, text_emb], 1)
    # gru1 runs on the transposed map => sequences along H ("vertical"), tsrn.py:391 / :423
    r = gru_block(p, prefix + ".gru1", r.transpose(-1, -2), explicit_rnn).transpose(-1, -2)
    return gru_block(p, prefix + ".gru2", x + r, explicit_rnn)


def stn_head(p, prefix: str, x: Tensor, training: bool) -> Tuple[Tensor, Tensor]:
    """STNHead.forward, model/stn_head.py:92-106 (activation='none')."""
    pools = [(2, 2), (2, 2), (2, 2), (2, 2), (1, 2), None]
    for i, pool in enumerate(pools):
        cp = f"{prefix}.stn_convnet.{2 * i}"
        x = F.conv2d(x, p[cp + ".0.weight"], p[cp + ".0.bias"], padding=1)
        x = F.relu(batch_norm(p, cp + ".1", x, training))
        if pool is not None:
            x = F.max_pool2d(x, pool, pool)
    x = x.reshape(x.shape[0], -1)
    feat = F.linear(x, p[prefix + ".stn_fc1.0.weight"], p[prefix + ".stn_fc1.0.bias"])
    feat = F.relu(batch_norm(p, prefix + ".stn_fc1.1", feat, training))
    ctrl = F.linear(0.1 * feat, p[prefix + ".stn_fc2.weight"], p[prefix + ".stn_fc2.bias"])
    return feat, ctrl.reshape(-1, ctrl.shape[1] // 2, 2)


def tps_transform(p, prefix: str, x: Tensor, ctrl: Tensor, out_hw: Tuple[int, int],
                  align_corners: bool = False) -> Tuple[Tensor, Tensor]:
    """TPSSpatialTransformer.forward, model/tps_spatial_transformer.py:97-112.  The reference calls
    F.grid_sample with defaults: bilinear, zeros padding, align_corners=False on torch >= 1.3."""
    n = ctrl.shape[0]
    y = torch.cat([ctrl, p[prefix + ".padding_matrix"].expand(n, 3, 2)], 1)
    mapping = torch.matmul(p[prefix + ".inverse_kernel"], y)
    src = torch.matmul(p[prefix + ".target_coordinate_repr"], mapping)
    grid = src.reshape(-1, out_hw[0], out_hw[1], 2).clamp(0, 1) * 2.0 - 1.0
    out = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    return out, src


def info_gen(p, prefix: str, t: Tensor, training: bool) -> Tensor:
    """InfoGen.forward, model/tsrn.py:100-108."""
    cfg = [((2, 2), (1, 1)), ((2, 2), (1, 1)), ((2, 2), (1, 1)), ((2, 1), (1, 0))]
    x = t
    for i, (stride, pad) in enumerate(cfg):
        x = F.conv_transpose2d(x, p[f"{prefix}.tconv{i + 1}.weight"], None, stride, pad)
        x = F.relu(batch_norm(p, f"{prefix}.bn{i + 1}", x, training))
    return x


def tsrn_forward(p, x: Tensor, text_emb: Optional[Tensor] = None, *, training: bool, stn: bool = True,
                 srb_nums: int = 5, text_prior: bool = False, scale_factor: int = 2,
                 grid_align_corners: bool = False, explicit_rnn: bool = False,
                 return_aux: bool = False):
    """TSRN.forward (model/tsrn.py:62-78) / TSRN_TL.forward (:178-215)."""
    aux = {}
    if stn and training:  # STN is bypassed in eval mode (tsrn.py:64 / :183)
        _, ctrl = stn_head(p, "stn_head", x, training)
        x, _ = tps_transform(p, "tps", x, ctrl, (x.shape[2], x.shape[3]), grid_align_corners)
        aux["ctrl"], aux["rectified"] = ctrl, x
    b1 = F.prelu(F.conv2d(x, p["block1.0.weight"], p["block1.0.bias"], padding=4), p["block1.1.weight"])
    temb = None
    if text_prior:
        if text_emb is None:  # tsrn.py:191-193
            text_emb = x.new_zeros(x.shape[0], p["infoGen.tconv1.weight"].shape[0], 1, 26)
        temb = info_gen(p, "infoGen", text_emb, training)
        temb = F.interpolate(temb, (x.shape[2], x.shape[3]), mode="bilinear", align_corners=True)
        aux["spatial_t_emb"] = temb
    cur = b1
    for i in range(srb_nums):
        cur = recurrent_residual_block(p, f"block{i + 2}", cur, training, temb, explicit_rnn)
    k = srb_nums + 2
    cur = F.conv2d(cur, p[f"block{k}.0.weight"], p[f"block{k}.0.bias"], padding=1)
    cur = batch_norm(p, f"block{k}.1", cur, training)
    cur = b1 + cur
    k += 1
    for u in range(int(math.log(scale_factor, 2))):  # UpsampleBLock, tsrn.py:464-477
        cur = F.conv2d(cur, p[f"block{k}.{u}.conv.weight"], p[f"block{k}.{u}.conv.bias"], padding=1)
        cur = mish(F.pixel_shuffle(cur, 2))
    u = int(math.log(scale_factor, 2))
    out = torch.tanh(F.conv2d(cur, p[f"block{k}.{u}.weight"], p[f"block{k}.{u}.bias"], padding=4))
    return (out, aux) if return_aux else out


def crnn_forward(p, gray: Tensor, *, training: bool, explicit_rnn: bool = False) -> Tensor:
    """CRNN.forward, model/crnn/crnn.py:74-90: (N,1,32,100) -> logits (T=26, N, 37)."""
    x = gray

    def cr(i, bn):
        nonlocal x
        pad = 1 if i < 6 else 0
        x = F.conv2d(x, p[f"cnn.conv{i}.weight"], p[f"cnn.conv{i}.bias"], padding=pad)
        if bn:
            x = batch_norm(p, f"cnn.batchnorm{i}", x, training)
        x = F.relu(x)

    cr(0, False); x = F.max_pool2d(x, 2, 2)
    cr(1, False); x = F.max_pool2d(x, 2, 2)
    cr(2, True); cr(3, False); x = F.max_pool2d(x, (2, 2), (2, 1), (0, 1))
    cr(4, True); cr(5, False); x = F.max_pool2d(x, (2, 2), (2, 1), (0, 1))
    cr(6, True)
    assert x.shape[2] == 1
    seq = x.squeeze(2).permute(2, 0, 1)  # (W, N, C)
    for j in range(2):  # BidirectionalLSTM, crnn.py:12-26
        r = lstm_bidir(seq, p, f"rnn.{j}.rnn", explicit_rnn)
        T, B, Hh = r.shape
        seq = F.linear(r.reshape(T * B, Hh), p[f"rnn.{j}.embedding.weight"], p[f"rnn.{j}.embedding.bias"])
        seq = seq.reshape(T, B, -1)
    return seq


def srcnn_forward(p, x: Tensor, scale_factor: int = 2) -> Tensor:
    """SRCNN.forward (STN=False), model/srcnn.py:132-145."""
    x = F.interpolate(x, scale_factor=scale_factor)  # nearest
    x = F.relu(F.conv2d(x, p["conv1.weight"], p["conv1.bias"], padding=4))
    x = F.relu(F.conv2d(x, p["conv2.weight"], p["conv2.bias"]))
    return F.conv2d(x, p["conv3.weight"], p["conv3.bias"], padding=2)


def parse_crnn_data(imgs: Tensor) -> Tensor:
    """TextBase.parse_crnn_data, interfaces/base.py:806-829: bicubic (A=-0.75, align_corners=False,
    no antialias) resize of RGB to (32,100), then luminance."""
    x = F.interpolate(imgs[:, :3], (32, 100), mode="bicubic")
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


# ---------------------------------------------------------------------------
# losses / metric
# ---------------------------------------------------------------------------
def gradient_map(x: Tensor) -> Tensor:
    """GradientPriorLoss.gradient_map, loss/image_loss.py:43-51 (zero-padded central differences)."""
    r = F.pad(x, (0, 1, 0, 0))[:, :, :, 1:]
    l = F.pad(x, (1, 0, 0, 0))[:, :, :, :-1]
    t = F.pad(x, (0, 0, 1, 0))[:, :, :-1, :]
    b = F.pad(x, (0, 0, 0, 1))[:, :, 1:, :]
    return torch.sqrt(((r - l) * 0.5) ** 2 + ((t - b) * 0.5) ** 2 + 1e-6)


def gradient_prior_loss(out: Tensor, target: Tensor) -> Tensor:
    """loss/image_loss.py:38-41 : L1 between gradient maps."""
    return (gradient_map(out) - gradient_map(target)).abs().mean()


def image_loss(out: Tensor, target: Tensor, gradient: bool = True, loss_weight=(1.0, 1e-4)) -> Tensor:
    """ImageLoss.forward, loss/image_loss.py:19-30: MSE over ALL channels, gradient loss over RGB."""
    loss = loss_weight[0] * F.mse_loss(out, target)
    if gradient:
        loss = loss + loss_weight[1] * gradient_prior_loss(out[:, :3], target[:, :3])
    return loss


def semantic_loss(pred: Tensor, gt: Tensor) -> Tensor:
    """SemanticLoss.forward, loss/semantic_loss.py:21-39: mean|gt-pred| + KLDivLoss('mean')."""
    margin = (gt - pred).abs().mean()
    q = gt + 1e-20
    kl = torch.where(q > 0, q * (torch.log(q) - torch.log(pred + 1e-20)), torch.zeros_like(q)).mean()
    return margin + kl


def calculate_psnr(a: Tensor, b: Tensor) -> Tensor:
    """utils/ssim_psnr.py:9-15 (whole batch, RGB only, x255)."""
    mse = ((a[:, :3] * 255 - b[:, :3] * 255) ** 2).mean()
    return 20 * torch.log10(255.0 / torch.sqrt(mse))


# ---------------------------------------------------------------------------
# optimiser pieces (interfaces/base.py:449-450, interfaces/super_resolution.py:419-424)
# ---------------------------------------------------------------------------
def clip_grad_norm_(grads: Sequence[Tensor], max_norm: float = 0.25) -> Tensor:
    """torch.nn.utils.clip_grad_norm_(..., 0.25) semantics: coef = max_norm/(total+1e-6), clamped to 1."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class AdamState:
    """Adam(lr=1e-3, betas=(0.5, 0.999), eps=1e-8, wd=0) -- interfaces/base.py:449-450."""

    def __init__(self, params: Sequence[Tensor], lr=1e-3, betas=(0.5, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.m = [torch.zeros_like(q) for q in self.params]
        self.v = [torch.zeros_like(q) for q in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads: Sequence[Tensor]):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for q, g, m, v in zip(self.params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            q.addcdiv_(m, denom, value=-self.lr / bc1)


# ---------------------------------------------------------------------------
# synthetic data and train steps (interfaces/super_resolution.py:295-424)
# ---------------------------------------------------------------------------
def synthetic_batch(n: int, seed: int, mask: bool = True, lr_hw=(16, 64), scale: int = 2):
    """HR = U[0,1) RGB (+ luminance-threshold mask channel, dataset/dataset.py:625-630 rule: 1 where the
    luminance is <= the per-image mean); LR = 2x average-pooled HR RGB with its own mask."""
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(n, 3, lr_hw[0] * scale, lr_hw[1] * scale, generator=g)
    lr = F.avg_pool2d(hr, scale)

    def add_mask(img):
        lum = 0.299 * img[:, 0:1] + 0.587 * img[:, 1:2] + 0.114 * img[:, 2:3]
        m = (lum <= lum.mean(dim=(1, 2, 3), keepdim=True)).float()
        return torch.cat([img, m], 1)

    if mask:
        hr, lr = add_mask(hr), add_mask(lr)
    return lr.contiguous(), hr.contiguous()


def tsrn_train_step(p, opt: AdamState, lr_img: Tensor, hr_img: Tensor, *, stn=True, srb_nums=5,
                    gradient=True, explicit_rnn=False, grid_align_corners=False):
    """Config C2: ``--arch tsrn`` branch, super_resolution.py:409-424:
    sr = model(lr); loss = ImageLoss(sr, hr).mean()*100; backward; clip(0.25); Adam."""
    sr = tsrn_forward(p, lr_img, training=True, stn=stn, srb_nums=srb_nums, explicit_rnn=explicit_rnn,
                      grid_align_corners=grid_align_corners)
    loss = image_loss(sr, hr_img, gradient).mean() * 100
    keys = trainable_keys(p)
    grads = list(torch.autograd.grad(loss, [p[k] for k in keys]))
    gnorm = clip_grad_norm_(grads, 0.25)
    opt.step(grads)
    return {"loss": loss.detach(), "grad_norm": gnorm, "sr": sr.detach(), "grads": dict(zip(keys, grads))}


def tpgsr_train_step(sr_params: List[dict], stu_params: List[dict], teacher: dict, opt: AdamState,
                     lr_img: Tensor, hr_img: Tensor, *, stu_iter=1, sr_share=True, tpg_share=False, stn=True,
                     srb_nums=5, gradient=True, explicit_rnn=False, grid_align_corners=False, tpg_forward=None, ssim_loss=False,
                     use_label=False, use_distill=True, labels=None):
    """Configs C3-C5: ``tsrn_tl_cascade`` branch, super_resolution.py:295-406 + :419-424.
    teacher(HR).detach -> per stage: student(prev image) -> softmax -> distill loss -> (N,37,1,26)
    -> zero the prior of samples [0, N//4) -> SR net -> image loss; sum; backward;
    clip each SR net to 0.25 (students are NOT clipped); one Adam over SR nets + students.
    tpg_forward(params, gray, training=...) -> (T, N, C) logits: the text-prior generator, `crnn_forward` (`--tpg CRNN`) unless given
    (`--tpg OPT`: oracle/opt_oracle.py:opt_forward; super_resolution.py:77-80 selects either for the same loop)."""
    if tpg_forward is None:
        tpg_forward = lambda q_, g_, training: crnn_forward(q_, g_, training=training, explicit_rnn=explicit_rnn)
    with torch.no_grad():
        t_logits = tpg_forward(teacher, parse_crnn_data(hr_img[:, :3]), training=False)
        q = F.softmax(t_logits, -1)
    cascade = lr_img
    loss_img = 0.0
    loss_distill = 0.0
    priors = []
    for i in range(stu_iter):
        stu = stu_params[0 if tpg_share else i]
        logits = tpg_forward(stu, parse_crnn_data(cascade[:, :3]), training=True)
        pv = F.softmax(logits, -1)                                   # (26, N, 37)
        prior = pv.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)  # (N, 37, 1, 26)
        if use_label:       # `--use_label`, super_resolution.py:347-366 (ctc_loss = torch.nn.CTCLoss(blank=0, reduction='none'), :40)
            label_vecs, weighted_mask, weighted_tics = labels
            text_sum = label_vecs.sum(1).squeeze(1)
            text_len = (text_sum > 0).float().sum(1).reshape(-1)
            predicted_length = torch.ones(logits.shape[1]) * logits.shape[0]
            fsup = F.ctc_loss(logits.log_softmax(2), weighted_mask.long(), predicted_length.long(), text_len.long(), blank=0, reduction="none")
            loss_distill = loss_distill + (fsup * weighted_tics.float()).mean()
        if use_distill:
            loss_distill = loss_distill + semantic_loss(pv, q) * 100
        drop = torch.ones(lr_img.shape[0])
        drop[: lr_img.shape[0] // 4] = 0.0
        prior = prior * drop.view(-1, 1, 1, 1)
        priors.append(pv.detach())
        srp = sr_params[0 if sr_share else i]
        cascade = tsrn_forward(srp, lr_img, prior, training=True, stn=stn, srb_nums=srb_nums, text_prior=True,
                               explicit_rnn=explicit_rnn, grid_align_corners=grid_align_corners)
        loss_img = loss_img + image_loss(cascade, hr_img, gradient).mean() * 100
        if ssim_loss:      # `--ssim_loss`, super_resolution.py:388-391: loss_ssim = (1 - ssim(cascade_images, images_hr).mean()) * 10.
            loss_img = loss_img + (1 - ssim(cascade, hr_img).mean()) * 10.
    loss = loss_img + loss_distill
    groups = [[(m, k) for k in trainable_keys(m)] for m in sr_params] + \
             [[(m, k) for k in trainable_keys(m)] for m in stu_params]
    flat = [m[k] for grp in groups for (m, k) in grp]
    grads = list(torch.autograd.grad(loss, flat, allow_unused=True))
    grads = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, flat)]
    ofs = 0
    gnorms = []
    for gi, grp in enumerate(groups):
        if gi < len(sr_params):
            gnorms.append(clip_grad_norm_(grads[ofs:ofs + len(grp)], 0.25))
        ofs += len(grp)
    opt.step(grads)
    return {"loss": loss.detach(), "loss_img": torch.as_tensor(loss_img).detach(),
            "loss_distill": torch.as_tensor(loss_distill).detach(), "grad_norms": gnorms,
            "sr": cascade.detach(), "priors": priors, "grads": grads}


COLLATE_D2A = "-0123456789abcdefghijklmnopqrstuvwxyz"


def collate_labels(label_strs):
    """The label half of alignCollate_realWTLAMask.__call__ (dataset/dataset.py:1255-1323; alphabet :1108-1116): per word (lower-cased, cut
    to 15 characters when longer than 14) the indices of its characters in "-0123456789a..z" (characters outside it are dropped) ->
    label_vecs (N, 37, 1, max_len) one-hot (max_len = the longest WORD), weighted_mask = all index lists concatenated (an empty list
    contributes one 0 and a one-hot on the blank), weighted_tics (N) = 1 for words with at least one label, else 0."""
    a2d = {ch: i for i, ch in enumerate(COLLATE_D2A)}
    alsize = len(COLLATE_D2A)
    max_len, batches, masks, tics = 0, [], [], []
    for word in label_strs:
        word = word.lower()
        if not (len(word) <= 1 or 1 < len(word) < 15):
            word = word[:15]
        label_list = [a2d[ch] for ch in word if ch in a2d]
        masks.extend(label_list if label_list else [0])
        max_len = max(max_len, len(word))
        if label_list:
            v = torch.zeros(len(label_list), alsize)
            v.scatter_(-1, torch.tensor(label_list)[:, None].long(), 1)
            tics.append(1)
        else:
            v = torch.zeros(1, alsize)
            v[0, 0] = 1.
            tics.append(0)
        batches.append(v)
    out = torch.zeros(len(label_strs), max_len, alsize)
    for i, v in enumerate(batches):
        out[i][:v.shape[0]] = v
    return out.unsqueeze(1).float().permute(0, 3, 1, 2), torch.tensor(masks).long(), torch.tensor(tics)


def srcnn_train_step(p, opt: AdamState, lr_img: Tensor, hr_img: Tensor):
    """Config C1: srcnn branch (3 channels, nn.MSELoss), super_resolution.py:409-424 + base.py:332-334."""
    sr = srcnn_forward(p, lr_img[:, :3])
    loss = F.mse_loss(sr, hr_img[:, :3]).mean() * 100
    keys = trainable_keys(p)
    grads = list(torch.autograd.grad(loss, [p[k] for k in keys]))
    gnorm = clip_grad_norm_(grads, 0.25)
    opt.step(grads)
    return {"loss": loss.detach(), "grad_norm": gnorm, "sr": sr.detach()}


# ---------------------------------------------------------------------------
# evaluation path (interfaces/super_resolution.py:540-900)
# ---------------------------------------------------------------------------
ALPHABET = "-0123456789abcdefghijklmnopqrstuvwxyz"


def get_string_crnn(outputs_: Tensor, alphabet: str = ALPHABET) -> List[str]:
    """utils/metrics.py:71-88: per sample arg-max over classes (first maximum), collapse repeats, drop the blank (index 0);
    a blank between two equal characters keeps both.  outputs_ is seq-first (T, N, C)."""
    idx = outputs_.permute(1, 0, 2).argmax(-1)       # torch.max / argmax return the first maximal index on CPU
    res = []
    for row in idx.tolist():
        s, last = "", -1
        for i in row:
            if i != last:
                if i != 0:
                    s += alphabet[i]
                    last = i
                else:
                    last = -1
        res.append(s)
    return res


def ssim(img1: Tensor, img2: Tensor, window_size: int = 11) -> Tensor:
    """utils/ssim_psnr.py:18-78 (SSIM(window_size=11, size_average=True)): Gaussian window sigma 1.5, zero padding, first 3 channels"""
    a, b = img1[:, :3], img2[:, :3]
    ch = a.shape[1]
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(ch, 1, window_size, window_size).contiguous()
    p = window_size // 2
    mu1, mu2 = F.conv2d(a, w, padding=p, groups=ch), F.conv2d(b, w, padding=p, groups=ch)
    s11 = F.conv2d(a * a, w, padding=p, groups=ch) - mu1 * mu1
    s22 = F.conv2d(b * b, w, padding=p, groups=ch) - mu2 * mu2
    s12 = F.conv2d(a * b, w, padding=p, groups=ch) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))).mean()


@torch.no_grad()
def tpgsr_eval_step(sr_params: List[dict], tpg_params: List[dict], recognizer: dict, lr_img: Tensor, hr_img: Tensor, *, stu_iter=1,
                    sr_share=True, tpg_share=False, srb_nums=5):
    """the cascade branch of TextSR.eval (interfaces/super_resolution.py:727-746 + :770-830): eval-mode networks (no STN, running
    BN statistics), per stage prior = softmax(TPG(parse_crnn_data(cascade))) -> SR; PSNR / SSIM of the last SR vs HR; strings
    of SR / LR / HR from the evaluation recogniser"""
    cascade = lr_img
    srs, priors = [], []
    for i in range(stu_iter):
        tpg = tpg_params[0 if tpg_share else i]
        pv = F.softmax(crnn_forward(tpg, parse_crnn_data(cascade[:, :3]), training=False), -1)
        prior = pv.permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
        cascade = tsrn_forward(sr_params[0 if sr_share else i], lr_img, prior, training=False, stn=True, srb_nums=srb_nums,
                               text_prior=True)
        srs.append(cascade)
        priors.append(pv)
    rec = lambda img: get_string_crnn(crnn_forward(recognizer, parse_crnn_data(img[:, :3]), training=False))
    return {"sr": srs, "priors": priors, "psnr": calculate_psnr(cascade, hr_img), "ssim": ssim(cascade, hr_img),
            "pred_sr": rec(cascade), "pred_lr": rec(lr_img), "pred_hr": rec(hr_img)}
