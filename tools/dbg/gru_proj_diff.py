#!/usr/bin/env python
"""The one-launch GruBlock forward at full batch (N 48, 16 x 64), affine + text-strip loader, H-axis scan: repeated launches against each other
and against projection + scan as two launches -- where (image, row, column, unit) do results differ, are outputs left unwritten?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from tpgsr_amd import kernels as K  # noqa: E402
from test_gru_proj_gpu import _case  # noqa: E402

DEV = "cuda"
N, H, W = 48, 16, 64
for loader, axis in (("affine+strip", 1), ("affine", 1)):
    Cin = 96 if loader == "affine+strip" else 64
    t, kw = _case(N, H, W, Cin, loader, seed=5)
    P = N * H * W
    geom = K.ConvGeom(N, H, W, Cin, 192)
    with K.conv_terms(2):
        K.make_bf_twin(t["wc"], 0)
        gi, h0, g0 = torch.empty(P, 192, device=DEV), torch.empty(P, 64, device=DEV), torch.empty(P, 256, device=DEV)
        K.conv_fwd(K.make_conv_args(geom, t["x"], t["wc"], gi, bias=t["bc"], **kw))
        K.bigru_fwd(gi, t["whh"], t["bhh"], N, H, W, axis, h0, g0)
        outs = []
        for rep in range(6):
            h, gt = torch.full((P, 64), float("nan"), device=DEV), torch.full((P, 256), float("nan"), device=DEV)
            K.bigru_proj_fwd(K.make_bigru_proj_args(K.make_conv_args(geom, t["x"], t["wc"], None, bias=t["bc"], **kw), t["whh"], t["bhh"], axis, h, gt))
            outs.append((h, gt))
    torch.cuda.synchronize()
    print(f"== {loader} axis {axis}")
    for i, (h, gt) in enumerate(outs):
        bad = ((h - h0).abs() > 1e-3) | torch.isnan(h)
        d0 = (h != outs[0][0]) & ~(torch.isnan(h) & torch.isnan(outs[0][0]))
        print(f"rep {i}: NaN in h {int(torch.isnan(h).sum())}, in gates {int(torch.isnan(gt).sum())}; |h - two launches| > 1e-3 at {int(bad.sum())} of {h.numel()}; differs from rep 0 at {int(d0.sum())}")
        if bad.any() and i < 2:
            idx = bad.nonzero()[:, 0].unique()
            n, r = idx // (H * W), idx % (H * W)
            cols = (r % W).unique()
            print("   images", n.unique().tolist()[:12], "rows", (r // W).unique().tolist(), "columns", cols.tolist()[:24], " col % 4:", (cols % 4).unique().tolist(),
                  "units", bad.nonzero()[:, 1].unique().tolist()[:12])
