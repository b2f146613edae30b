"""Synthetic weights and batches for benchmarking (no checkpoint or dataset ships with the reference: SURVEY.md section 8d).

Product-side on purpose: `bench.py` builds its networks and inputs from here and never needs the CPU oracle to do so (the oracle's
`recipe_state_dict` / `synthetic_batch` are what the PARITY tests share with the reference; these helpers follow the same recipe --
tests/test_synthetic_cpu.py checks that -- but nothing here is used to check anything)."""
import math

import numpy as np
import torch

from ..model.nn_params import BatchNormParams, ConvTranspose2dParams, PReLUParams

_TPS_BUFFERS = ("inverse_kernel", "padding_matrix", "target_coordinate_repr", "target_control_points")


def _identity_ctrl_points(k: int, margin: float = 0.01) -> np.ndarray:
    """model/stn_head.py:73-90 (init_stn): k points along the top edge, k along the bottom edge"""
    xs = np.linspace(margin, 1.0 - margin, k)
    return np.concatenate([np.stack([xs, np.full(k, margin)], 1), np.stack([xs, np.full(k, 1.0 - margin)], 1)], 0).reshape(-1)


def init_by_recipe(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Fill every parameter / BatchNorm buffer of `module` in state_dict key order from numpy.random.default_rng(seed) with a
    per-kind scale ("pseudo-trained" weights: activations and gradients of realistic size in every layer, BatchNorm away from the
    identity, the STN head near -- not at -- the identity transform).  TPS constant buffers are left as constructed."""
    rng = np.random.default_rng(seed)
    kinds = {}
    for mname, m in module.named_modules():
        pre = mname + "." if mname else ""
        if isinstance(m, BatchNormParams):
            kinds.update({pre + "weight": "bn_w", pre + "bias": "bn_b", pre + "running_mean": "bn_rm", pre + "running_var": "bn_rv"})
        elif isinstance(m, PReLUParams):
            kinds[pre + "weight"] = "prelu"
        elif isinstance(m, ConvTranspose2dParams):
            kinds[pre + "weight"] = "tconv_w"
    sd = module.state_dict()
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        last = name.split(".")[-1]
        if last == "num_batches_tracked" or last in _TPS_BUFFERS:
            out[name] = t
            continue
        kind = kinds.get(name)
        if name.endswith("stn_fc2.weight"):
            a = rng.standard_normal(shape) * 0.02
        elif name.endswith("stn_fc2.bias"):
            a = _identity_ctrl_points(shape[0] // 4) + rng.standard_normal(shape) * 0.03
        elif kind == "bn_w":
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind in ("bn_b", "bn_rm"):
            a = 0.1 * rng.standard_normal(shape)
        elif kind == "bn_rv":
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "prelu":
            a = np.full(shape, 0.25) + rng.standard_normal(shape) * 0.02
        elif "bias" in last:
            a = rng.standard_normal(shape) * 0.1
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            if kind == "tconv_w":          # ConvTranspose2d weight is (Cin, Cout, kh, kw)
                fan_in = shape[0] * shape[2] * shape[3] // 2
            a = rng.standard_normal(shape) * (1.0 / math.sqrt(max(fan_in, 1)))
        out[name] = torch.tensor(np.asarray(a), dtype=torch.float32).reshape(shape)
    module.load_state_dict(out, strict=True)
    return module


def synthetic_batch(n: int, seed: int, lr_hw=(16, 64), scale: int = 2, mask: bool = True):
    """HR = U[0,1) RGB + the luminance-threshold mask channel of dataset/dataset.py:625-630 (1 where the luminance is <=
    the per-image mean); LR = `scale`x average-pooled HR RGB with its own mask.  -> (lr (n,4,h,w), hr (n,4,h*scale,w*scale)) on the CPU"""
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(n, 3, lr_hw[0] * scale, lr_hw[1] * scale, generator=g)
    lr = torch.nn.functional.avg_pool2d(hr, scale)

    def add_mask(img):
        lum = 0.299 * img[:, 0:1] + 0.587 * img[:, 1:2] + 0.114 * img[:, 2:3]
        return torch.cat([img, (lum <= lum.mean(dim=(1, 2, 3), keepdim=True)).float()], 1)

    if mask:
        hr, lr = add_mask(hr), add_mask(lr)
    return lr.contiguous(), hr.contiguous()
